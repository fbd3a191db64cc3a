// Host side of libdmme_hip: the UNet execution plan (layer graph, parameter table,
// packed-weight layout, workspace layout, launch sequence) and the extern "C" API.
#include "plan.h"

namespace dmme {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int debug_route(const char* key, int dflt) {
    const char* e = getenv("DMME_DEBUG_ROUTE");
    if (!e || !*e) return dflt;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        if (len >= kl && !strncmp(p, key, kl) && (len == kl || p[kl] == '=')) return len == kl ? 1 : atoi(p + kl + 1);
        if (!end) break;
        p = end + 1;
    }
    return dflt;
}

}  // namespace dmme

using namespace dmme;

namespace dmme {

struct Node {
    int kind;  // 0 res, 1 down, 2 up
    std::string prefix;
    int cin, cout;
    bool attn;
    // parameter indices
    int gn1w = -1, gn1b = -1, c1w = -1, c1b = -1, tw = -1, tb = -1, gn2w = -1, gn2b = -1, c2w = -1, c2b = -1;
    int rw = -1, rb = -1, anw = -1, anb = -1, qw = -1, qb = -1, pw = -1, pb = -1;
    int cw = -1, cb = -1;  // down / up conv
    int tproj_col = 0;
};

struct Builder {
    dmme_plan* P;
    int64_t ref_cursor = 0;

    int add_param(const std::string& name, std::initializer_list<int64_t> shape, bool is_buffer, bool as_f32,
                  int cout, int cin, int taps) {
        Param p;
        p.name = name;
        p.ndim = (int)shape.size();
        int i = 0;
        for (auto s : shape) p.shape[i++] = s;
        p.ref_off = ref_cursor;
        p.is_buffer = is_buffer;
        p.as_f32 = as_f32;
        p.cout = cout;
        p.cin = cin;
        p.taps = taps;
        ref_cursor += p.numel();
        P->params.push_back(p);
        return (int)P->params.size() - 1;
    }
    void conv(const std::string& p, int ci, int co, int k, int& w, int& b) {
        w = add_param(p + ".weight", {co, ci, k, k}, false, false, co, ci, k * k);
        b = add_param(p + ".bias", {co}, false, true, 1, co, 1);
    }
    void lin(const std::string& p, int ci, int co, int& w, int& b) {
        w = add_param(p + ".weight", {co, ci}, false, false, co, ci, 1);
        b = add_param(p + ".bias", {co}, false, true, 1, co, 1);
    }
    void gn(const std::string& p, int c, int& w, int& b) {
        w = add_param(p + ".weight", {c}, false, true, 1, c, 1);
        b = add_param(p + ".bias", {c}, false, true, 1, c, 1);
    }
    void res_params(Node& n) {
        const dmme_unet_cfg& c = P->cfg;
        const std::string& p = n.prefix;
        if (c.arch == DMME_ARCH_IDDPM) {
            // iddpm.ResBlock.__init__ registration order (models/iddpm.py:85-104): conv1 (built with p = 0: conv at index 2),
            // norm, condition (Linear emb -> 2 c_out), conv2 = norm_act_drop_conv(...)[1:] (the slice keeps the keys 1..3)
            gn(p + ".conv1.0", n.cin, n.gn1w, n.gn1b);
            conv(p + ".conv1.2", n.cin, n.cout, 3, n.c1w, n.c1b);
            gn(p + ".norm", n.cout, n.gn2w, n.gn2b);
            lin(p + ".condition.0", c.emb_dim, 2 * n.cout, n.tw, n.tb);
            conv(p + (c.dropout > 0 ? ".conv2.3" : ".conv2.2"), n.cout, n.cout, 3, n.c2w, n.c2b);
            if (n.cin != n.cout) conv(p + ".residual", n.cin, n.cout, 1, n.rw, n.rb);
            if (n.attn) {
                gn(p + ".attention.norm", n.cout, n.anw, n.anb);
                conv(p + ".attention.qkv_proj", n.cout, 3 * n.cout, 1, n.qw, n.qb);
                conv(p + ".attention.proj", n.cout, n.cout, 1, n.pw, n.pb);
            }
            return;
        }
        gn(p + ".conv1.0", n.cin, n.gn1w, n.gn1b);
        conv(p + ".conv1.2", n.cin, n.cout, 3, n.c1w, n.c1b);
        lin(p + ".condition.0", c.emb_dim, n.cout, n.tw, n.tb);
        gn(p + ".conv2.0", n.cout, n.gn2w, n.gn2b);
        // norm_act_drop_conv (models/ddpm.py:25-35): conv index 3 with Dropout2d, 2 without
        conv(p + (c.dropout > 0 ? ".conv2.3" : ".conv2.2"), n.cout, n.cout, 3, n.c2w, n.c2b);
        if (n.cin != n.cout) conv(p + ".residual", n.cin, n.cout, 1, n.rw, n.rb);
        if (n.attn) {
            gn(p + ".attention.norm", n.cout, n.anw, n.anb);
            conv(p + ".attention.qkv_proj", n.cout, 3 * n.cout, 1, n.qw, n.qb);
            conv(p + ".attention.proj", n.cout, n.cout, 1, n.pw, n.pb);
        }
    }
};

bool in_list(const int* v, int n, int x) {
    for (int i = 0; i < n; ++i)
        if (v[i] == x) return true;
    return false;
}

int build_plan(dmme_plan* P) {
    const dmme_unet_cfg& c = P->cfg;
    const int nb = c.num_blocks, nd = c.num_depths;
    const int B = P->B;
    const int64_t es = (int64_t)dtype_size(P->dtype);

    // ---- layer graph: UNet.__init__ (models/ddpm.py:203-279) ----
    std::vector<int> chans;
    chans.push_back(c.channels_per_depth[0]);
    for (int d = 0; d < nd; ++d)
        for (int b = 0; b < nb; ++b) chans.push_back(c.channels_per_depth[d]);
    const int L = (int)chans.size();  // len(channels)
    auto is_cut = [&](int layer_num) {  // layer_num in downsample_layers = {nb*i : 1 <= i < nd}
        return layer_num > 0 && layer_num % nb == 0 && layer_num / nb <= nd - 1;
    };
    auto has_attn = [&](int depth) { return in_list(c.attention_depths, c.num_attention_depths, depth); };

    std::vector<Node> down, mid, up;
    int depth = 1;
    for (int i = 0; i + 1 < L; ++i) {
        Node n{0, "down_layers." + std::to_string(down.size()), chans[i], chans[i + 1], has_attn(depth)};
        down.push_back(n);
        if (is_cut(i + 1)) {
            Node d{1, "down_layers." + std::to_string(down.size()), chans[i + 1], chans[i + 1], false};
            down.push_back(d);
            ++depth;
        }
    }
    // the reference's `if down_layers[-1] == len(channels) - 1` (:242) compares a module
    // with an int and is never true: the up path starts with ResBlocks, never an UpSample
    depth = nd;
    for (int i = 0; i + 1 < L; ++i) {
        const int ci = chans[L - 1 - i], co = chans[L - 2 - i];
        const bool at = has_attn(depth);
        const int layer_num = L - 1 - i;
        up.push_back(Node{0, "up_layers." + std::to_string(up.size()), 2 * ci, co, at});
        if (is_cut(layer_num - 1)) {
            up.push_back(Node{0, "up_layers." + std::to_string(up.size()), 2 * co, co, at});
            up.push_back(Node{2, "up_layers." + std::to_string(up.size()), co, co, false});
            --depth;
        }
    }
    up.push_back(Node{0, "up_layers." + std::to_string(up.size()), 2 * chans[0], chans[0], has_attn(1)});
    const int top = chans[L - 1];
    mid.push_back(Node{0, "middle_layers.0", top, top, true});
    mid.push_back(Node{0, "middle_layers.1", top, top, false});
    if (c.arch == DMME_ARCH_IDDPM)  // MultiHeadAttention asserts dim % num_heads == 0 (models/iddpm.py:26)
        for (auto* seq : {&down, &up, &mid})
            for (auto& n : *seq)
                DMME_REQUIRE(!n.attn || n.cout % c.num_heads == 0, DMME_ERR_INVALID, "%s: attention width %d not divisible by num_heads=%d",
                             n.prefix.c_str(), n.cout, c.num_heads);

    DMME_REQUIRE(!(P->mix && has_attn(1)), DMME_ERR_UNSUPPORTED, "precision fp16r32: attention at the full-resolution level (attention_depths contains 1) is not supported");

    // ---- parameter table in nn.Module registration order ----
    Builder bld{P};
    const int half = c.pos_dim / 2;
    P->freqs_param = bld.add_param("condition.0.embeddings", {1, half}, true, true, 1, half, 1);
    int l1w, l1b, l2w, l2b, icw, icb, ogw, ogb, ocw, ocb;
    bld.lin("condition.1", c.pos_dim, c.emb_dim, l1w, l1b);
    bld.lin("condition.3", c.emb_dim, c.emb_dim, l2w, l2b);
    P->p_l1w = l1w; P->p_l1b = l1b; P->p_l2w = l2w; P->p_l2b = l2b;
    bld.conv("input_conv", c.in_channels, chans[0], 3, icw, icb);
    for (auto* seq : {&down, &up, &mid})
        for (auto& n : *seq) {
            if (n.kind == 0)
                bld.res_params(n);
            else if (n.kind == 1)
                bld.conv(n.prefix, n.cin, n.cout, 3, n.cw, n.cb);
            else
                bld.conv(n.prefix + ".conv", n.cin, n.cout, 3, n.cw, n.cb);
        }
    bld.gn("output_conv.0", chans[0], ogw, ogb);
    P->out_channels = c.arch == DMME_ARCH_IDDPM ? 2 * c.in_channels : c.in_channels;
    bld.conv("output_conv.2", chans[0], P->out_channels, 3, ocw, ocb);
    P->ref_numel = bld.ref_cursor;

    // ---- packed layout: the per-block time projections form one [sumCout][emb] matrix ----
    int tcols = 0;
    for (auto* seq : {&down, &up, &mid})
        for (auto& n : *seq)
            if (n.kind == 0) {
                const int width = c.arch == DMME_ARCH_IDDPM ? 2 * n.cout : n.cout;  // (shift | scale) for the scale-shift blocks
                n.tproj_col = tcols;
                tcols += width;
                P->tblocks.push_back({n.tw, n.tb, n.tproj_col, width});
            }
    P->tproj_cols = tcols;
    auto layout_packed = [&]() {
        int64_t cur = 0;
        P->tproj_w_off = cur;
        cur = align_up(cur + (int64_t)tcols * c.emb_dim * es, 256);
        P->tproj_b_off = cur;
        cur = align_up(cur + (int64_t)tcols * 4, 256);
        std::unordered_map<int, const Node*> tw_owner, tb_owner;
        for (auto* seq : {&down, &up, &mid})
            for (auto& n : *seq)
                if (n.kind == 0) {
                    tw_owner[n.tw] = &n;
                    tb_owner[n.tb] = &n;
                }
        for (int i = 0; i < (int)P->params.size(); ++i) {
            Param& p = P->params[i];
            if (tw_owner.count(i)) {
                p.packed_off = P->tproj_w_off + (int64_t)tw_owner[i]->tproj_col * c.emb_dim * es;
            } else if (tb_owner.count(i)) {
                p.packed_off = P->tproj_b_off + (int64_t)tb_owner[i]->tproj_col * 4;
            } else {
                p.packed_off = cur;
                cur = align_up(cur + p.numel() * ((p.as_f32 || p.pack_code >= 3) ? 4 : es), 256);
            }
        }
        P->packed_bytes = cur;
    };
    {   // transposed + tap-flipped conv weights for the data-gradient convolutions
        int64_t bc = 0;
        for (Param& p : P->params)
            if (p.ndim == 4) {
                p.packed_bwd_off = bc;
                bc = align_up(bc + p.numel() * es, 256);
            }
        P->packed_bwd_bytes = bc;
    }

    // ---- workspace + op list ----
    int64_t ws = 0;
    auto ws_alloc = [&](int64_t bytes) {
        const int64_t o = ws;
        ws = align_up(ws + bytes, 256);
        return o;
    };
    auto is_top = [&](int H_, int W_) { return P->mix && H_ == P->H && W_ == P->W; };
    auto new_tensor = [&](int C, int H, int W, int force16 = 0) {
        Tensor t;
        t.C = C;
        t.H = H;
        t.W = W;
        t.f32 = is_top(H, W) && !force16;
        t.off = ws_alloc((int64_t)B * H * W * C * (t.f32 ? 4 : es));
        P->tensors.push_back(t);
        return (int)P->tensors.size() - 1;
    };
    P->ws_tsin = ws_alloc((int64_t)B * c.pos_dim * 4);
    P->ws_th1 = ws_alloc((int64_t)B * c.emb_dim * 4);
    P->ws_temb = ws_alloc((int64_t)B * c.emb_dim * 4);
    P->ws_tz1 = ws_alloc((int64_t)B * c.emb_dim * 4);
    P->ws_tz2 = ws_alloc((int64_t)B * c.emb_dim * 4);
    P->ws_tproj = ws_alloc((int64_t)B * tcols * 4);

    std::vector<Op>& ops = P->ops;
    {
        Op o{};
        o.kind = OP_SINUS;
        ops.push_back(o);
        Op a{};
        a.kind = OP_LINEAR;
        a.lin_in = P->ws_tsin; a.lin_out = P->ws_th1; a.lin_K = c.pos_dim; a.lin_N = c.emb_dim;
        a.lin_pre = P->ws_tz1;
        a.lin_w = l1w; a.lin_b = l1b; a.lin_silu = 1;
        ops.push_back(a);
        Op b{};
        b.kind = OP_LINEAR;
        b.lin_in = P->ws_th1; b.lin_out = P->ws_temb; b.lin_K = c.emb_dim; b.lin_N = c.emb_dim;
        b.lin_pre = P->ws_tz2;
        b.lin_w = l2w; b.lin_b = l2b; b.lin_silu = 1;
        ops.push_back(b);
        Op d{};
        d.kind = OP_LINEAR;  // all per-block projections at once (w/b = -1: concatenated region)
        d.lin_in = P->ws_temb; d.lin_out = P->ws_tproj; d.lin_K = c.emb_dim; d.lin_N = tcols;
        d.lin_w = -1; d.lin_b = -1; d.lin_silu = 0;
        ops.push_back(d);
    }
    size_t gn_part_max = 0;
    auto emit_gn = [&](int s1, int s2, int gw, int gb) {
        Op o{};
        o.kind = OP_GN;
        o.gn_src1 = s1; o.gn_src2 = s2; o.gn_gamma = gw; o.gn_beta = gb;
        const Tensor& t1 = P->tensors[s1];
        const int C = t1.C + (s2 >= 0 ? P->tensors[s2].C : 0);
        o.gn_scale = ws_alloc((int64_t)B * C * 4);
        o.gn_shift = ws_alloc((int64_t)B * C * 4);
        o.gn_mr = ws_alloc((int64_t)B * c.num_groups * 2 * 4);
        const size_t part = gn_fast_scratch_floats(B, t1.H * t1.W, C, c.num_groups);
        if (part > gn_part_max) gn_part_max = part;
        ops.push_back(o);
        return (int)ops.size() - 1;
    };
    int64_t dmask_cursor = 0;
    int cur_t, H = P->H, W = P->W;
    {
        Op o{};
        o.kind = OP_CONV;
        o.src1 = -2; o.w = icw; o.b = icb; o.taps = 9;
        o.dst = new_tensor(chans[0], H, W);
        if (P->mix) {  // fp32 in, fp32 out: the fp32 instance of the input conv (its 27-deep products are fp32 MFMAs anyway)
            o.route_f32 = 1;
            P->params[icw].pack_code = 3;
        }
        ops.push_back(o);
        cur_t = o.dst;
        P->named["input_conv"] = cur_t;
    }
    std::vector<int> skips{cur_t};

    auto emit_res = [&](const Node& n, int x1, int x2) {
        const Tensor tx = P->tensors[x1];
        const int h = tx.H, w = tx.W;
        const bool iddpm = P->cfg.arch == DMME_ARCH_IDDPM;
        const int g1 = emit_gn(x1, x2, n.gn1w, n.gn1b);
        Op c1{};
        c1.kind = OP_CONV;
        c1.src1 = x1; c1.src2 = x2; c1.w = n.c1w; c1.b = n.c1b; c1.gn = g1; c1.pro_silu = 1;
        c1.tproj_col = iddpm ? -1 : n.tproj_col; c1.taps = 9;  // DDPM: h += Linear(t_emb) in the epilogue (models/ddpm.py:129)
        c1.dst = new_tensor(n.cout, h, w);
        const bool top = is_top(h, w);
        if (top) {
            c1.mix = 1;
            P->params[n.c1w].pack_code = 4;
        }
        ops.push_back(c1);
        const int hmid = c1.dst;
        const int g2 = emit_gn(hmid, -1, n.gn2w, n.gn2b);
        if (iddpm) {  // h = norm(h) * (scale + 1) + shift (models/iddpm.py:117-118)
            ops[g2].gn_mod_col = n.tproj_col;
            ops[g2].gn_mod_C = n.cout;
        }
        int r1 = x1, r2 = x2;
        if (n.cin != n.cout) {
            Op rc{};
            rc.kind = OP_CONV;
            rc.src1 = x1; rc.src2 = x2; rc.w = n.rw; rc.b = n.rb; rc.taps = 1;
            rc.dst = new_tensor(n.cout, h, w);
            if (top) {
                rc.mix = 4;
                P->params[n.rw].pack_code = 4;
            }
            ops.push_back(rc);
            r1 = rc.dst;
            r2 = -1;
        }
        Op c2{};
        c2.kind = OP_CONV;
        c2.src1 = hmid; c2.w = n.c2w; c2.b = n.c2b; c2.gn = g2; c2.pro_silu = 1; c2.taps = 9;
        if (P->cfg.dropout > 0) c2.dmask_off = dmask_cursor;
        dmask_cursor += (int64_t)B * n.cout;
        c2.res1 = r1; c2.res2 = r2;
        c2.dst = new_tensor(n.cout, h, w);
        if (top) {
            c2.mix = 1;
            P->params[n.c2w].pack_code = 4;
        }
        ops.push_back(c2);
        int out = c2.dst;
        if (n.attn) {
            const int g3 = emit_gn(out, -1, n.anw, n.anb);
            Op q{};
            q.kind = OP_CONV;
            q.src1 = out; q.w = n.qw; q.b = n.qb; q.gn = g3; q.taps = 1;
            q.dst = new_tensor(3 * n.cout, h, w);
            ops.push_back(q);
            Op at{};
            at.kind = OP_ATTN;
            at.at_qkv = q.dst;
            at.at_heads = iddpm ? P->cfg.num_heads : 1;
            at.at_out = new_tensor(n.cout, h, w);
            at.at_lse = ws_alloc((int64_t)B * at.at_heads * h * w * 4);
            ops.push_back(at);
            Op pr{};
            pr.kind = OP_CONV;
            pr.src1 = at.at_out; pr.w = n.pw; pr.b = n.pb; pr.taps = 1; pr.res1 = out;
            pr.dst = new_tensor(n.cout, h, w);
            ops.push_back(pr);
            out = pr.dst;
        }
        P->named[n.prefix] = out;
        return out;
    };

    for (auto& n : down) {
        if (n.kind == 0) {
            cur_t = emit_res(n, cur_t, -1);
        } else {
            DMME_REQUIRE(H % 2 == 0 && W % 2 == 0, DMME_ERR_UNSUPPORTED, "odd resolution %dx%d at a DownSample", H, W);
            if (P->tensors[cur_t].f32) {  // leaving the fp32 level: the stride-2 conv reads a 16-bit copy
                Op cs{};
                cs.kind = OP_CAST;
                cs.cast_src = cur_t;
                cs.cast_dst = new_tensor(P->tensors[cur_t].C, H, W, 1);
                ops.push_back(cs);
                cur_t = cs.cast_dst;
            }
            Op o{};
            o.kind = OP_CONV;
            o.src1 = cur_t; o.w = n.cw; o.b = n.cb; o.taps = 9; o.stride = 2;
            H /= 2; W /= 2;
            o.dst = new_tensor(n.cout, H, W);
            ops.push_back(o);
            cur_t = o.dst;
            P->named[n.prefix] = cur_t;
        }
        skips.push_back(cur_t);
    }
    for (auto& n : mid) cur_t = emit_res(n, cur_t, -1);
    for (auto& n : up) {
        if (n.kind == 0) {
            DMME_REQUIRE(!skips.empty(), DMME_ERR_INVALID, "skip stack underflow");
            const int sk = skips.back();
            skips.pop_back();
            const Tensor &a = P->tensors[cur_t], &b = P->tensors[sk];
            DMME_REQUIRE(a.C + b.C == n.cin && a.H == b.H && a.W == b.W, DMME_ERR_INVALID,
                         "%s: concat %d+%d channels does not match conv input %d", n.prefix.c_str(), a.C, b.C, n.cin);
            cur_t = emit_res(n, cur_t, sk);  // torch.cat([x, skip]) : x first (:310)
        } else {
            Op o{};
            o.kind = OP_CONV;
            o.src1 = cur_t; o.w = n.cw; o.b = n.cb; o.taps = 9; o.up = 1;
            H *= 2; W *= 2;
            o.dst = new_tensor(n.cout, H, W);
            if (is_top(H, W)) {  // entering the fp32 level from the 16-bit one: split passes on a 16-bit source
                o.mix = P->tensors[cur_t].f32 ? 1 : 2;
                P->params[n.cw].pack_code = 4;
            }
            ops.push_back(o);
            cur_t = o.dst;
            P->named[n.prefix] = cur_t;
        }
    }
    {
        const int g = emit_gn(cur_t, -1, ogw, ogb);
        Op o{};
        o.kind = OP_CONV;
        o.src1 = cur_t; o.w = ocw; o.b = ocb; o.gn = g; o.pro_silu = 1; o.taps = 9; o.dst = -2;
        if (P->mix && P->tensors[cur_t].f32) {
            o.mix = 3;
            P->params[ocw].pack_code = 4;
        }
        ops.push_back(o);
    }
    P->dropmask_numel = dmask_cursor;
    if (P->mix) {
        // which split-pass 3x3 convs run TWO passes: bit k of the mask = the k-th such conv in op order (default below; DMME_DEBUG_ROUTE
        // r32_2pass=<mask> overrides, r32_2pass=0 = three passes everywhere)
        const int mask = debug_route("r32_2pass", kR32TwoPassDefault);
        int k = 0;
        for (Op& o : ops)
            if (o.kind == OP_CONV && (o.mix == 1 || o.mix == 2)) {
                o.mix2 = (mask >> k) & 1;
                ++k;
            }
    }
    layout_packed();  // (after the op list: a mixed plan's convs choose their filter layouts above)
    P->ws_gnpart = ws_alloc((int64_t)(gn_part_max ? gn_part_max : 1) * 4);
    {   // up to 4 partial images of the widest small-map conv (either direction: its data gradient has Cin outputs)
        int64_t mx = 0;
        for (const Op& o : ops) {
            if (o.kind != OP_CONV || o.taps != 9 || o.src1 < 0 || o.dst < 0) continue;
            const Tensor &ti = P->tensors[o.src1], &to = P->tensors[o.dst];
            if (to.H * to.W > 64 || o.stride != 1 || o.up) continue;
            const int cin = ti.C + (o.src2 >= 0 ? P->tensors[o.src2].C : 0);
            const int64_t v = (int64_t)B * to.H * to.W * (cin > to.C ? cin : to.C);
            if (v > mx) mx = v;
        }
        P->splitk_floats = 4 * mx;
        P->ws_splitk = ws_alloc(P->splitk_floats * 4 + 16);
    }
    P->ws_bytes = ws;
    P->n_launches = (int)ops.size();

    // ---- backward workspace: one gradient buffer per forward tensor + temporaries ----
    {
        int64_t bw = 0;
        auto balloc = [&](int64_t bytes) {
            const int64_t o = bw;
            bw = align_up(bw + bytes, 256);
            return o;
        };
        int64_t tmp_max = 0, att_max = 0;
        int cmax = c.in_channels;
        for (const Tensor& t : P->tensors) {
            P->gt_off.push_back(balloc((int64_t)B * t.H * t.W * t.C * es));
            if (t.C > cmax) cmax = t.C;
        }
        // A ResBlock's 1x1 residual conv feeds nothing but the residual input of conv2: the gradient of its output IS the gradient of
        // the block's output - the two tensors share one gradient buffer instead of a copy launch per block
        if (!debug_route("no_res_alias")) {
            std::vector<int> uses(P->tensors.size(), 0), producer(P->tensors.size(), -1);
            for (int oi = 0; oi < (int)ops.size(); ++oi) {
                const Op& o = ops[oi];
                for (int id : {o.src1, o.src2, o.res1, o.res2, o.gn_src1, o.gn_src2, o.at_qkv})
                    if (id >= 0) ++uses[id];
                if (o.kind == OP_CONV && o.dst >= 0) producer[o.dst] = oi;
            }
            for (Op& o : ops) {
                if (o.kind != OP_CONV || o.res1 < 0 || o.res2 >= 0 || o.dst < 0) continue;
                const int r = o.res1;
                if (uses[r] != 1 || producer[r] < 0 || ops[producer[r]].kind != OP_CONV || P->tensors[r].C != P->tensors[o.dst].C) continue;
                P->gt_off[r] = P->gt_off[o.dst];
                o.res_alias = 1;
            }
        }
        for (const Op& o : ops) {
            if (o.kind == OP_CONV && o.src1 >= 0) {
                const Tensor& t1 = P->tensors[o.src1];
                const int Cin = t1.C + (o.src2 >= 0 ? P->tensors[o.src2].C : 0);
                const int64_t up = o.up ? 4 : 1;
                const int64_t b = (int64_t)B * t1.H * t1.W * up * Cin * es;
                if (b > tmp_max) tmp_max = b;
            }
            if (o.kind == OP_ATTN) {
                const Tensor& q = P->tensors[o.at_qkv];
                const int64_t b = (int64_t)B * o.at_heads * q.H * q.W * q.H * q.W * 4;
                if (b > att_max) att_max = b;
            }
        }
        {   // accumulation scratch, one contiguous region cleared by a single memset per backward:
            // packed-layout weight-gradient image, per-conv column sums, per-GroupNorm channel sums
            P->bws_zero = bw;
            int64_t wfl = 0;
            for (Param& p : P->params)
                if (p.ndim == 4) {
                    p.wp_off = wfl;
                    wfl += (p.numel() + 63) / 64 * 64;
                }
            P->bws_wimage = balloc(wfl * 4);
            for (Op& o : P->ops) {
                if (o.kind != OP_CONV) continue;
                o.b_rowsum = balloc((int64_t)B * P->params[o.w].cout * 4);
            }
            P->bws_zpage = balloc(256);  // a page of zeros: the padding rows of the DMA-fed weight gradient
            P->bws_zero_bytes = bw - P->bws_zero;
            for (Op& o : P->ops) {  // GroupNorm channel sums, one partial row per pixel chunk (written whole: outside the cleared region)
                if (o.kind != OP_CONV || o.gn < 0) continue;
                const Op& gop = P->ops[o.gn];
                const Tensor& t1 = P->tensors[gop.gn_src1];
                const int C = t1.C + (gop.gn_src2 >= 0 ? P->tensors[gop.gn_src2].C : 0);
                o.b_ab = balloc((int64_t)gn_bwd_fast_chunks(P->dtype, t1.H * t1.W, C) * B * C * 2 * 4);
                o.b_gnrows = balloc((int64_t)2 * B * C * 4);
            }
            P->bws_gnS = balloc((int64_t)B * c.num_groups * 2 * 4);
        }
        P->bws_tmp = balloc(tmp_max);
        P->bws_dy = balloc((int64_t)B * P->H * P->W * P->out_channels * es);
        P->bws_rowsum = balloc((int64_t)B * cmax * 3 * 4);  // qkv convs have 3*C outputs
        P->bws_dtproj = balloc((int64_t)B * tcols * 4);
        P->bws_dtemb = balloc((int64_t)B * c.emb_dim * 4);
        P->bws_dh1 = balloc((int64_t)B * c.emb_dim * 4);
        P->bws_z = balloc((int64_t)B * c.emb_dim * 4);
        P->bws_wT = balloc((int64_t)(tcols > c.emb_dim ? tcols : c.emb_dim) * c.emb_dim * es);
        P->bws_attP = balloc(att_max);
        P->bws_attdS = balloc(att_max);
        P->bws_bytes = bw;
    }
    return DMME_OK;
}

// re-pack items: (cout range) x (cin range) sub-blocks of at most 8192 elements (the LDS tile of pack_table_kernel).
// wide_co: prefer long cout runs (the transposed data-gradient layout writes runs of couts), else long cin runs.
static void push_pack_items(const Param& p, int64_t dst_off, int as_f32, bool wide_co, std::vector<PackItem>& items) {
    const int LIMIT = 8192;
    int nco, nci;
    if (as_f32 == 1) {
        nci = p.cin;
        nco = LIMIT / (p.cin * p.taps);
        if (nco < 1) nco = 1;
    } else if (wide_co) {
        nco = p.cout < 64 ? p.cout : 64;
        nci = LIMIT / (nco * p.taps);
        if (nci > p.cin) nci = p.cin;
        if (nci >= 4) nci &= ~3;  // whole 16-byte vectors per source run (pack_table_kernel's vector loads)
        if (nci < 1) nci = 1;
    } else {
        nci = p.cin;
        if (nci * p.taps > LIMIT) nci = LIMIT / p.taps;
        nco = LIMIT / (nci * p.taps);
        if (nco < 1) nco = 1;
    }
    if (as_f32 != 1 && nco > 256) nco = 256;  // the odd LDS pitch adds one float per cout: keep that within the tile's slack
    for (int r0 = 0; r0 < p.cout; r0 += nco)
        for (int c0 = 0; c0 < p.cin; c0 += nci) {
            PackItem it;
            it.src_off = p.ref_off;
            it.dst_off = dst_off;
            it.cout = p.cout;
            it.cin = p.cin;
            it.taps = p.taps;
            it.row0 = r0;
            it.rows = p.cout - r0 < nco ? p.cout - r0 : nco;
            it.ci0 = c0;
            it.nci = p.cin - c0 < nci ? p.cin - c0 : nci;
            it.as_f32 = as_f32;
            items.push_back(it);
        }
}

int build_pack_items(dmme_plan* P, std::vector<PackItem>& items) {
    for (const Param& p : P->params) push_pack_items(p, p.packed_off, p.pack_code >= 0 ? p.pack_code : p.as_f32 ? 1 : 0, false, items);
    return DMME_OK;
}

int build_pack_items_bwd(dmme_plan* P, std::vector<PackItem>& items) {
    for (const Param& p : P->params)
        if (p.packed_bwd_off >= 0) push_pack_items(p, p.packed_bwd_off, 2, true, items);
    return DMME_OK;
}

int build_unpack_items(dmme_plan* P, std::vector<PackItem>& items) {
    const int64_t CHUNK = 16384;
    for (const Param& p : P->params) {
        if (p.wp_off < 0) continue;
        const int64_t row = (int64_t)p.cin * p.taps;
        int64_t rows_per = CHUNK / row;
        if (rows_per < 1) rows_per = 1;
        for (int64_t r0 = 0; r0 < p.cout; r0 += rows_per) {
            PackItem it;
            it.src_off = p.ref_off;  // destination: reference-layout gradient (float offset)
            it.dst_off = p.wp_off;   // source: packed-layout image (float offset)
            it.cout = p.cout;
            it.cin = p.cin;
            it.taps = p.taps;
            it.row0 = (int32_t)r0;
            it.rows = (int32_t)((p.cout - r0) < rows_per ? (p.cout - r0) : rows_per);
            it.ci0 = 0;
            it.nci = p.cin;
            it.as_f32 = 0;
            items.push_back(it);
        }
    }
    return DMME_OK;
}

// ---- which kernel runs a convolution -----------------------------------------------------------------------------------------------
// ONE table for the whole dispatch (the *_supported / *_pick functions below and in the kernel files implement exactly this; times are
// per launch at the benchmark configuration - default UNet, batch 128, bf16 - from profiles/r04_sample_b128_bf16_kernel_stats.csv and
// bench.py's event-bracketed forward; "B <= 32" rows from profiles/r04_bench_n1.json: small_batch):
//
//   shape class (16-bit plans)                               kernel                                   launches/step   us/launch
//   3x3 s1, 32x32 and 16x16 maps, Cin % 128 == 0, >= 256     conv3x3_ws2_kernel<11, T, 256>               18            66-72
//     256-pixel tiles of one image (the dominant kernel)       wave-specialised, persistent
//   ... the same where only 128-pixel tiles fill the chip    conv3x3_ws2_kernel<7, T, 128>                 4            41
//     (128-cout layers of the 16x16 level; 32x32 at B = 32)
//   3x3 s1 on 8x8 / 4x4 maps, DDPM blocks, <= 2 iterations   lvl_engine_kernel (plan_lvl.hip: a whole     3        131 / 181
//     per workgroup                                            level per launch; includes its 1x1 convs,
//                                                              norms and the 4x4 attention)
//   3x3 s1, few output pixels (small batches; IDDPM small    conv3x3_kw_kernel<NI, RING, DENSE, BM>       -           13-18
//     maps; 8x8 / 4x4 with DMME_NO_LVL)                        K split over the four waves
//   3x3 s2 (DownSample), 3x3 with fused 2x upsampling,       conv3x3_pipe_kernel<T, 64, 64, 3 | 9, UA>     4           36-45
//     everything the rows above decline                        four-wave software pipeline
//   1x1, K = 128 / 256, >= 128 tiles of 128 pixels           conv1x1_as_kernel<KCH, RES>                  15           14-25
//     (qkv, proj; the blocks' residual convs of the 32x32 /    activations stationary in registers
//      16x16 levels only with DMME_DEBUG_ROUTE=no_rseg: they
//      are a second K segment of conv2's launch, assign_rseg)
//   1x1 otherwise (K = 384 / 512, small maps, small batches) conv1x1_pipe_kernel<T, BM, BN>                2           20-30
//   output conv (<= 7 couts, NCHW fp32 out)                  conv_out_thin_kernel<NT, T>                   1            15
//   input conv (NCHW fp32 in, <= 4 channels)                 conv_in_mfma_kernel<T, CT> (generic file)     1            31
//   fp32 plans / precision="bf16x3" (fp32 tensors)           conv3x3_pipe / conv1x1_pipe <float[, ACC3]>;  -             -
//                                                              the 3-cout output conv: conv_mfma_kernel
//   precision="fp16r32", full-resolution level (ConvArgs::mix)  conv3x3_ws2_kernel<11, f16, 256, SPLIT>   11          120-240
//                                                              conv_out_thin_kernel<.., SPLIT>; its 1x1 residual
//                                                              convs and input conv on the fp32-tensor kernels
//   anything else (odd channel counts: the tiny test net)    conv_generic_kernel                           -             -
//
// A/B switches (read when a plan is built or a launch is dispatched; they select among these kernels, never a CPU path): a dozen
// product switches as environment variables of their own - DMME_NO_LVL, DMME_NO_WS, DMME_NO_KW, DMME_NO_CONV1X1_AS, DMME_NO_CONV_THIN,
// DMME_NO_FUSED_GN, DMME_NO_GN_IN, DMME_NO_GN_DIRECT, DMME_NO_PREACT, DMME_NO_ATTN_FULL, DMME_NO_WGRAD_GROUP, DMME_NO_GN_BWD_REGS,
// DMME_NO_GRAD_BUCKETS - and every experiment / comparison route as a key of DMME_DEBUG_ROUTE="key[=int],..." (debug_route()).
int run_any_conv(int dtype, const ConvArgs& a, hipStream_t s) {
    if (conv_out_thin_supported(dtype, a)) return launch_conv_out_thin(a, s);
    if (conv1x1_pipe_supported(dtype, a)) return launch_conv1x1_pipe(dtype, a, s);
    if (conv_pipe_supported(dtype, a)) return launch_conv_pipe(dtype, a, s);
    if (conv_mfma_supported(dtype, a)) return launch_conv_mfma(dtype, a, s);
    return launch_conv_generic(dtype, a, s);
}

// fill the device-side descriptor of a conv op
// fwd: the forward launch (a conv whose GroupNorm pre-activated its input reads that tensor and applies nothing); the backward
// passes false and sees the raw sources with their scale / shift / mask, which is what it differentiates through
void fill_conv(const dmme_plan* P, const Op& o, const char* packed, const float* x, float* y, char* ws,
               const float* drop_masks, int nt, ConvArgs& a, bool fwd) {
    const int64_t es = (int64_t)dtype_size(P->dtype);
    (void)es;
    a.x3 = P->x3 || o.route_f32;
    a.f16 = P->dtype == DMME_F16 && !o.route_f32;
    a.mix = o.mix;
    a.mix2 = o.mix2;
    a.N = P->B;
    if (o.src1 == -2) {
        a.src1 = x;
        a.in_nchw = 1;
        a.C1 = P->cfg.in_channels;
        a.Hin = P->H;
        a.Win = P->W;
    } else {
        const Tensor& t = P->tensors[o.src1];
        a.src1 = ws + t.off;
        a.C1 = t.C;
        a.Hin = t.H;
        a.Win = t.W;
    }
    if (o.src2 >= 0) {
        a.src2 = ws + P->tensors[o.src2].off;
        a.C2 = P->tensors[o.src2].C;
    }
    const Param& w = P->params[o.w];
    a.w = packed + w.packed_off;
    a.bias = (const float*)(packed + P->params[o.b].packed_off);
    a.Cout = w.cout;
    a.taps = o.taps;
    a.stride = o.stride;
    a.up = o.up;
    const int Hv = o.up ? 2 * a.Hin : a.Hin, Wv = o.up ? 2 * a.Win : a.Win;
    a.Hout = Hv / o.stride;
    a.Wout = Wv / o.stride;
    if (o.gn >= 0) {
        a.scale = (const float*)(ws + P->ops[o.gn].gn_scale);
        a.shift = (const float*)(ws + P->ops[o.gn].gn_shift);
        const Op& g = P->ops[o.gn];
        if (fwd && g.gn_in_consumer) {
            const Tensor& t1 = P->tensors[g.gn_src1];
            const Tensor* t2 = g.gn_src2 >= 0 ? &P->tensors[g.gn_src2] : nullptr;
            a.has_gni = 1;
            a.gni.p1 = (const float*)(ws + t1.stats_off);
            a.gni.t1 = t1.stats_tiles;
            a.gni.cnt1 = t1.stats_cnt;
            a.gni.C1 = t1.C;
            a.gni.p2 = t2 ? (const float*)(ws + t2->stats_off) : nullptr;
            a.gni.t2 = t2 ? t2->stats_tiles : 0;
            a.gni.cnt2 = t2 ? t2->stats_cnt : 0;
            a.gni.C2 = t2 ? t2->C : 0;
            a.gni.groups = P->cfg.num_groups;
            a.gni.gamma = (const float*)(packed + P->params[g.gn_gamma].packed_off);
            a.gni.beta = (const float*)(packed + P->params[g.gn_beta].packed_off);
            a.gni.eps = 1e-5f;
            a.gni.mean_rstd = (float*)(ws + g.gn_mr);
            if (g.gn_mod_col >= 0) {  // scale-shift conditioning: (shift | scale) columns of the batched time projection
                a.gni.t_shift = (const float*)(ws + P->ws_tproj) + g.gn_mod_col;
                a.gni.t_scale = a.gni.t_shift + g.gn_mod_C;
                a.gni.t_ld = P->tproj_cols;
                a.gni.nt = nt;
            }
        }
    }
    a.pro_silu = o.pro_silu;
    a.out_silu = o.out_silu;
    if (o.dmask_off >= 0 && drop_masks) a.dmask = drop_masks + o.dmask_off;
    if (fwd && o.use_act) {
        a.src1 = ws + P->ops[o.gn].gn_act;
        a.C1 = a.C1 + a.C2;
        a.src2 = nullptr;
        a.C2 = 0;
        a.scale = a.shift = a.dmask = nullptr;
        a.pro_silu = 0;
    }
    if (o.tproj_col >= 0) {
        a.tproj = (const float*)(ws + P->ws_tproj) + o.tproj_col;
        a.tproj_ld = P->tproj_cols;
        a.nt = nt;
    }
    if (fwd && o.rseg >= 0) {  // the block's residual conv runs inside this launch: no residual tensor
        const Op& rc = P->ops[o.rseg];
        const Tensor& r1 = P->tensors[rc.src1];
        a.r_src1 = ws + r1.off;
        a.r_C1 = r1.C;
        if (rc.src2 >= 0) {
            a.r_src2 = ws + P->tensors[rc.src2].off;
            a.r_C2 = P->tensors[rc.src2].C;
        }
        a.r_w = packed + P->params[rc.w].packed_off;
        a.r_bias = (const float*)(packed + P->params[rc.b].packed_off);
    } else if (o.res1 >= 0) {
        a.res1 = ws + P->tensors[o.res1].off;
        a.R1 = P->tensors[o.res1].C;
        if (o.res2 >= 0) a.res2 = ws + P->tensors[o.res2].off;
    }
    if (P->splitk_floats > 0 && ws) {
        a.splitk = (float*)(ws + P->ws_splitk);
        a.splitk_cap = P->splitk_floats;
    }
    if (fwd && o.gd_n > 0) {
        a.n_gno = o.gd_n;
        a.gn_eps = 1e-5f;
        a.gn_cg = P->tensors[o.dst].C / P->cfg.num_groups;
        for (int k = 0; k < o.gd_n; ++k) {
            const Op& g = P->ops[o.gd_gn[k]];
            GnOut& G = a.gno[k];
            G.gamma = (const float*)(packed + P->params[g.gn_gamma].packed_off);
            G.beta = (const float*)(packed + P->params[g.gn_beta].packed_off);
            G.scale = (float*)(ws + g.gn_scale);
            G.shift = (float*)(ws + g.gn_shift);
            G.mean_rstd = (float*)(ws + g.gn_mr);
            G.C = P->tensors[g.gn_src1].C + (g.gn_src2 >= 0 ? P->tensors[g.gn_src2].C : 0);
            G.cg = G.C / P->cfg.num_groups;
            G.c_off = o.gd_coff[k];
            if (k == o.gd_act) {
                const Op& cv = P->ops[g.gn_consumer];
                a.act = ws + g.gn_act;
                a.act_k = k;
                a.act_silu = cv.pro_silu;
                a.act_dmask = (cv.dmask_off >= 0 && drop_masks) ? drop_masks + cv.dmask_off : nullptr;
            }
        }
    }
    if (o.dst == -2) {
        a.dst = y;
        a.out_nchw = 1;
    } else {
        const Tensor& td = P->tensors[o.dst];
        a.dst = ws + td.off;
        if (td.stats_off >= 0) {
            a.gn_part = (float*)(ws + td.stats_off);
            a.gn_cg = td.C / P->cfg.num_groups;
            a.gn_tiles = td.stats_tiles;
        }
    }
}

// can this GroupNorm be finalised from the partials its producers emitted?
bool gn_from_parts(const dmme_plan* P, const Op& o) {
    if (o.gn_force_small) return false;
    const Tensor& t1 = P->tensors[o.gn_src1];
    if (t1.stats_off < 0) return false;
    const int G = P->cfg.num_groups;
    int C = t1.C;
    if (o.gn_src2 >= 0) {
        const Tensor& t2 = P->tensors[o.gn_src2];
        if (t2.stats_off < 0) return false;
        C += t2.C;
        const int cg = C / G;
        if (t1.C % cg || cg % (t1.C / G) || cg % (t2.C / G)) return false;
    }
    return true;
}

// decide which conv outputs carry fused GroupNorm partials (needs the launch-time tile choice) and give them room
void assign_stats(dmme_plan* P) {
    std::vector<char> wanted(P->tensors.size(), 0);
    for (const Op& o : P->ops)
        if (o.kind == OP_GN && o.lvl < 0) {  // (norms inside a level run are finished by the engine)
            wanted[o.gn_src1] = 1;
            if (o.gn_src2 >= 0) wanted[o.gn_src2] = 1;
        }
    const int G = P->cfg.num_groups;
    int64_t ws = P->ws_bytes;
    for (const Op& o : P->ops) {
        if (o.kind != OP_CONV || o.dst < 0 || !wanted[o.dst] || o.lvl >= 0) continue;
        Tensor& td = P->tensors[o.dst];
        if (td.C % G) continue;
        ConvArgs a{};
        fill_conv(P, o, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a);
        a.nt = o.tproj_col >= 0 ? P->B : 0;  // training adds one time-embedding row per image
        int tiles = 0, px = 0;
        if (!conv_stats_query(conv_dt(P, o), a, td.C / G, &tiles, &px)) continue;
        td.stats_tiles = tiles;
        td.stats_cnt = px * (td.C / G);
        td.stats_off = ws;
        ws = align_up(ws + (int64_t)P->B * tiles * G * 2 * 4, 256);
    }
    P->ws_bytes = ws;
}

// Small maps again, from the producer's side: a conv whose tile is 64 pixels of WHOLE images (8x8, 4x4 maps) sees every value of an
// (image, group) in one workgroup, so its epilogue can finish the norms that consume its output - scale / shift / {mean, rstd} rows
// and, for a single-source norm, the consumer's pre-activated input - and the norm itself is no launch at all
// (conv_epilogue_store_direct).  A tensor goes this way only if ALL the norms reading it do (else it would need partials as well),
// and a norm only if all its sources do.
void assign_direct(dmme_plan* P) {
    const int nT = (int)P->tensors.size(), nO = (int)P->ops.size();
    const int G = P->cfg.num_groups;
    std::vector<int> producer(nT, -1);
    for (int oi = 0; oi < nO; ++oi)
        if (P->ops[oi].kind == OP_CONV && P->ops[oi].dst >= 0 && P->ops[oi].lvl < 0) producer[P->ops[oi].dst] = oi;
    struct Use { int gn, coff; };
    std::vector<std::vector<Use>> uses(nT);
    for (int oi = 0; oi < nO; ++oi) {
        const Op& g = P->ops[oi];
        if (g.kind != OP_GN || g.lvl >= 0) continue;
        uses[g.gn_src1].push_back({oi, 0});
        if (g.gn_src2 >= 0) uses[g.gn_src2].push_back({oi, P->tensors[g.gn_src1].C});
    }
    std::vector<char> elig(nT, 0);
    for (int t = 0; t < nT; ++t) {
        if (producer[t] < 0 || uses[t].empty() || uses[t].size() > 2) continue;
        const Op& o = P->ops[producer[t]];
        ConvArgs a{};
        fill_conv(P, o, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a);
        a.nt = o.tproj_col >= 0 ? P->B : 0;
        a.gn_part = nullptr;
        a.n_gno = (int)uses[t].size();
        bool ok = true;
        int cgs[2] = {0, 0};
        for (size_t k = 0; k < uses[t].size(); ++k) {
            const Op& g = P->ops[uses[t][k].gn];
            const int C = P->tensors[g.gn_src1].C + (g.gn_src2 >= 0 ? P->tensors[g.gn_src2].C : 0);
            if (g.gn_mod_col >= 0 || C % G || uses[t][k].coff % (C / G)) ok = false;
            cgs[k] = C / G;
        }
        a.gn_cg = P->tensors[t].C / G;
        if (ok && conv_gn_direct_query(conv_dt(P, o), a, cgs, (int)uses[t].size())) elig[t] = 1;
        else if (ok && P->tensors[t].C % G == 0 && conv_gn_direct_ws_query(conv_dt(P, o), a, cgs, (int)uses[t].size())) elig[t] = 2;  // no act output
    }
    for (bool changed = true; changed;) {  // a norm needs all its sources, a tensor all its norms
        changed = false;
        for (int t = 0; t < nT; ++t) {
            if (!elig[t]) continue;
            for (const Use& u : uses[t]) {
                const Op& g = P->ops[u.gn];
                if (!elig[g.gn_src1] || (g.gn_src2 >= 0 && !elig[g.gn_src2])) {
                    elig[t] = 0;
                    changed = true;
                    break;
                }
            }
        }
    }
    int64_t ws = P->ws_bytes;
    const int64_t es = (int64_t)dtype_size(P->dtype);
    for (int t = 0; t < nT; ++t) {
        if (!elig[t]) continue;
        Op& o = P->ops[producer[t]];
        P->tensors[t].stats_off = -1;  // no partials: nothing merges them
        o.gd_n = (int)uses[t].size();
        for (int k = 0; k < o.gd_n; ++k) {
            o.gd_gn[k] = uses[t][k].gn;
            o.gd_coff[k] = uses[t][k].coff;
            P->ops[uses[t][k].gn].gn_direct = 1;
        }
    }
    if (getenv("DMME_NO_PREACT")) return;
    // single-source direct norms: the producer also writes the consumer's pre-activated input
    for (int ci = 0; ci < nO; ++ci) {
        Op& cv = P->ops[ci];
        if (cv.kind != OP_CONV || cv.gn < 0 || cv.src1 < 0 || cv.up || cv.stride != 1 || cv.lvl >= 0) continue;
        Op& g = P->ops[cv.gn];
        if (!g.gn_direct || g.gn_src2 >= 0 || g.gn_act >= 0 || g.gn_src1 != cv.src1 || cv.src2 >= 0) continue;
        Op& pr = P->ops[producer[g.gn_src1]];
        if (pr.gd_act >= 0 || elig[g.gn_src1] != 1) continue;  // one pre-activated output per producer; not from the two-pass kernel
        for (int k = 0; k < pr.gd_n; ++k)
            if (pr.gd_gn[k] == cv.gn) pr.gd_act = k;
        if (pr.gd_act < 0) continue;
        const Tensor& t1 = P->tensors[g.gn_src1];
        g.gn_act = ws;
        g.gn_consumer = ci;
        cv.use_act = 1;
        ws = align_up(ws + (int64_t)P->B * t1.H * t1.W * t1.C * es, 256);
    }
    P->ws_bytes = ws;
}

// Small maps: the GroupNorms that run as one workgroup per image (maps of at most 64 pixels: their producers tile several images
// together and cannot fuse the statistics) also write the consumer conv's pre-activated input, once per element.  Not for the
// scale-shift blocks of the IDDPM UNet (their per-(n, c) affine is modulated after the norm kernel).
void assign_preact(dmme_plan* P) {
    int64_t ws = P->ws_bytes;
    const int64_t es = (int64_t)dtype_size(P->dtype);
    for (int ci = 0; ci < (int)P->ops.size(); ++ci) {
        Op& cv = P->ops[ci];
        if (cv.kind != OP_CONV || cv.gn < 0 || cv.src1 < 0 || cv.up || cv.stride != 1 || cv.lvl >= 0) continue;
        Op& g = P->ops[cv.gn];
        constexpr bool over_parts = false;  // (the whole-image norm kernel over tensors whose producers left partials: measured neutral, below)
        if (g.gn_mod_col >= 0 || g.gn_act >= 0 || g.gn_direct) continue;
        const Tensor& t1 = P->tensors[g.gn_src1];
        const int C2 = g.gn_src2 >= 0 ? P->tensors[g.gn_src2].C : 0;
        if (g.gn_src1 != cv.src1 || g.gn_src2 != cv.src2) continue;
        if (!gn_small_act_supported(P->dtype, t1.H * t1.W, t1.C, C2, P->cfg.num_groups)) continue;
        // 8x8 maps: one image per 64-pixel tile, so the producers did leave partials and the norm is a 5 us finalize launch.  Taking
        // the whole-image kernel there instead (DMME_PREACT_PARTS=1) was measured neutral: the consumers drop 34.9 -> 29.1 us, the
        // three-pass norm kernel costs 12 us instead of 5.5 (step 2.999 vs 3.016 ms) - off by default
        if (gn_from_parts(P, g)) {
            if (!over_parts) continue;
            g.gn_force_small = 1;
        }
        g.gn_act = ws;
        g.gn_consumer = ci;
        cv.use_act = 1;
        ws = align_up(ws + (int64_t)P->B * t1.H * t1.W * (t1.C + C2) * es, 256);
    }
    P->ws_bytes = ws;
}

// GroupNorm statistics folded with the affine into per-(n, c) scale / shift for the consuming conv's prologue
// Norms finished by their CONSUMER: a conv on the wave-specialised kernel whose norm's statistics are the producers' partials merges
// them in its parameter fill (ws_fill_par_gni) - no finalize launch between the two convolutions.
void assign_gn_in(dmme_plan* P) {
    for (Op& cv : P->ops) {
        if (cv.kind != OP_CONV || cv.gn < 0 || cv.use_act || cv.lvl >= 0) continue;
        Op& g = P->ops[cv.gn];
        if (g.gn_direct || g.gn_in_consumer || g.gn_act >= 0 || !gn_from_parts(P, g)) continue;
        if (g.gn_src1 != cv.src1 || g.gn_src2 != cv.src2) continue;
        const Tensor& t1 = P->tensors[g.gn_src1];
        // partials per consumer group: tiles x (producer groups per consumer group), at most 64 (one or two batches of loads in the
        // fill); 64x64 maps have hundreds - the batched finalize kernel's job
        const int G = P->cfg.num_groups, Cn = t1.C + (g.gn_src2 >= 0 ? P->tensors[g.gn_src2].C : 0);
        bool fits = t1.stats_tiles * ((Cn / G) / (t1.C / G)) <= 64;
        if (g.gn_src2 >= 0) {
            const Tensor& t2 = P->tensors[g.gn_src2];
            fits = fits && t2.stats_tiles * ((Cn / G) / (t2.C / G)) <= 64;
        }
        if (!fits) continue;
        ConvArgs a{};
        fill_conv(P, cv, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a);
        a.nt = cv.tproj_col >= 0 ? P->B : 0;
        if (!conv_gn_in_query(conv_dt(P, cv), a)) continue;
        g.gn_in_consumer = 1;
    }
}

// The ResBlocks' 1x1 residual convs (models/ddpm.py:108-111: `h + residual(x)` where the channel count changes) whose conv2 runs on the
// wave-specialised 3x3 kernel's 256-pixel form become a second K segment of that launch (conv_pipe.hip, RSEG): at the benchmark batch the
// seven of them on the 32x32 / 16x16 levels were 0.17 ms of launches that mostly moved bytes (raw input in, residual tensor out, and
// in again in conv2's epilogue).  Forward only: the backward pass never reads a residual tensor.  DMME_DEBUG_ROUTE=no_rseg: off.
void assign_rseg(dmme_plan* P) {
    if (debug_route("no_rseg") || !is16(P->dtype) || P->mix || P->x3) return;
    std::vector<int> uses(P->tensors.size(), 0), producer(P->tensors.size(), -1);
    for (int oi = 0; oi < (int)P->ops.size(); ++oi) {
        const Op& o = P->ops[oi];
        for (int id : {o.src1, o.src2, o.res1, o.res2, o.gn_src1, o.gn_src2, o.at_qkv, o.cast_src})
            if (id >= 0) ++uses[id];
        if (o.kind == OP_CONV && o.dst >= 0) producer[o.dst] = oi;
    }
    for (int ci = 0; ci < (int)P->ops.size(); ++ci) {
        Op& c2 = P->ops[ci];
        if (c2.kind != OP_CONV || c2.taps != 9 || c2.lvl >= 0 || c2.res1 < 0 || c2.res2 >= 0 || c2.dst < 0 || c2.mix || c2.route_f32) continue;
        const int ri = producer[c2.res1];
        if (ri < 0 || uses[c2.res1] != 1) continue;
        Op& rc = P->ops[ri];
        if (rc.kind != OP_CONV || rc.taps != 1 || rc.lvl >= 0 || rc.gn >= 0 || rc.pro_silu || rc.out_silu || rc.dmask_off >= 0 || rc.tproj_col >= 0 || rc.res1 >= 0 ||
            rc.up || rc.stride != 1 || rc.src1 < 0 || rc.gd_n > 0 || rc.mix || rc.route_f32)
            continue;
        for (const auto& kv : P->named)  // (a tensor somebody can ask for by module name stays a tensor)
            if (kv.second == c2.res1) goto next;
        {
            c2.rseg = ri;
            ConvArgs a{};
            fill_conv(P, c2, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a, true);
            if (!conv_pipe_rseg_supported(P->dtype, a)) {
                c2.rseg = -1;
                continue;
            }
            rc.fused_away = 1;
        }
    next:;
    }
}

// The attention blocks of the 16x16 maps (models/ddpm.py:66-75: `x + proj(attention(norm(x)))`): where the whole-row attention kernel
// serves the block (256 keys, single head, a launch that fills the chip), its proj 1x1 conv, bias, residual add and the next norm's
// partial statistics run inside that launch (attn_mfma.hip, AttnProj) - at the benchmark batch five launches of 8-15 us and the context
// tensor's round trip.  Forward only.  DMME_DEBUG_ROUTE=no_attn_proj: off.
void assign_attn_proj(dmme_plan* P) {
    if (P->x3) return;
    for (int ai = 0; ai + 1 < (int)P->ops.size(); ++ai) {
        Op& at = P->ops[ai];
        if (at.kind != OP_ATTN || at.lvl >= 0 || at.at_heads != 1) continue;
        Op& pr = P->ops[ai + 1];
        if (pr.kind != OP_CONV || pr.src1 != at.at_out || pr.src2 >= 0 || pr.taps != 1 || pr.lvl >= 0 || pr.gn >= 0 || pr.pro_silu || pr.out_silu || pr.dmask_off >= 0 ||
            pr.tproj_col >= 0 || pr.res1 < 0 || pr.res2 >= 0 || pr.up || pr.stride != 1 || pr.gd_n > 0 || pr.mix || pr.route_f32 || pr.dst < 0 || pr.fused_away || pr.rseg >= 0)
            continue;
        const Tensor &q = P->tensors[at.at_qkv], &td = P->tensors[pr.dst];
        if (q.f32 || td.f32 || P->tensors[pr.res1].f32 || P->tensors[at.at_out].f32) continue;
        const int S = q.H * q.W, C = q.C / 3;
        if (td.C != C || P->tensors[pr.res1].C != C) continue;
        const bool stats = td.stats_off >= 0;
        if (!attn_proj_fusable(P->dtype, P->B, S, C, stats ? td.C / P->cfg.num_groups : 0, stats ? td.stats_tiles : 0)) continue;
        at.at_proj = ai + 1;
        pr.fused_away = 2;
    }
}

int run_gn(const dmme_plan* P, const Op& o, const char* pk, char* ws, int nt, const float* drop_masks, hipStream_t s) {
    if (o.gn_direct || o.gn_in_consumer) return DMME_OK;  // its producers (its consumer) wrote scale / shift / {mean, rstd} (and act)
    // scale-shift conditioning (iddpm.ResBlock): (shift | scale) columns of the batched time projection
    const float* tsh = o.gn_mod_col >= 0 ? (const float*)(ws + P->ws_tproj) + o.gn_mod_col : nullptr;
    const float* tsc = tsh ? tsh + o.gn_mod_C : nullptr;
    int rc;
    const Tensor& t1 = P->tensors[o.gn_src1];
    const void* s1 = ws + t1.off;
    const void* s2 = o.gn_src2 >= 0 ? ws + P->tensors[o.gn_src2].off : nullptr;
    const int C2 = o.gn_src2 >= 0 ? P->tensors[o.gn_src2].C : 0;
    const float* gam = (const float*)(pk + P->params[o.gn_gamma].packed_off);
    const float* bet = (const float*)(pk + P->params[o.gn_beta].packed_off);
    float* sc = (float*)(ws + o.gn_scale);
    float* sh = (float*)(ws + o.gn_shift);
    if (gn_from_parts(P, o)) {
        const Tensor* t2 = o.gn_src2 >= 0 ? &P->tensors[o.gn_src2] : nullptr;
        return launch_gn_finalize_parts((const float*)(ws + t1.stats_off), t1.stats_tiles, t1.stats_cnt, t1.C,
                                        t2 ? (const float*)(ws + t2->stats_off) : nullptr, t2 ? t2->stats_tiles : 0,
                                        t2 ? t2->stats_cnt : 0, C2, P->B, P->cfg.num_groups, gam, bet, 1e-5f, sc, sh,
                                        (float*)(ws + o.gn_mr), tsh, tsc, P->tproj_cols, nt, s);
    }
    const int gdt = t1.f32 ? DMME_F32 : P->dtype;  // (a mixed plan's fp32 level; concatenated sources share a level)
    if (gn_fast_supported(gdt, P->B, t1.H * t1.W, t1.C, C2, P->cfg.num_groups)) {
        void* act = nullptr;
        int act_silu = 0;
        const float* dm = nullptr;
        if (o.gn_act >= 0) {
            const Op& cv = P->ops[o.gn_consumer];
            act = ws + o.gn_act;
            act_silu = cv.pro_silu;
            if (cv.dmask_off >= 0 && drop_masks) dm = drop_masks + cv.dmask_off;
        }
        rc = launch_gn_fast(gdt, s1, s2, P->B, t1.H * t1.W, t1.C, C2, P->cfg.num_groups, gam, bet, 1e-5f, sc, sh,
                            (float*)(ws + o.gn_mr), (float*)(ws + P->ws_gnpart), s, act, act_silu, dm);
    }
    else
        rc = launch_gn_generic(gdt, s1, s2, P->B, t1.H * t1.W, t1.C, C2, P->cfg.num_groups, gam, bet, 1e-5f, sc, sh,
                               (float*)(ws + o.gn_mr), s);
    if (rc != DMME_OK || !tsh) return rc;
    return launch_gn_modulate(sc, sh, tsh, tsc, P->tproj_cols, nt, P->B, o.gn_mod_C, s);
}

// keep_ctx: a backward pass may follow this forward (tensors only the backward reads are written: the fused attention block's context)
int run_op(const dmme_plan* P, const Op& o, const char* pk, const float* x, const int64_t* t, int nt, float* y,
           char* ws, const float* drop_masks, hipStream_t s, bool keep_ctx) {
    if (o.lvl >= 0) return o.lvl_first ? run_level(P, P->lvl_runs[o.lvl], pk, ws, nt, drop_masks, s, keep_ctx) : DMME_OK;
    if (o.fused_away) return DMME_OK;  // a residual 1x1 conv that runs inside its block's conv2 (assign_rseg)
    switch (o.kind) {
        case OP_SINUS:
            return launch_time_sinusoid(t, nt, (const float*)(pk + P->params[P->freqs_param].packed_off),
                                        P->cfg.pos_dim / 2, (float*)(ws + P->ws_tsin), s);
        case OP_LINEAR: {
            const char* w = o.lin_w >= 0 ? pk + P->params[o.lin_w].packed_off : pk + P->tproj_w_off;
            const float* b = (const float*)(o.lin_b >= 0 ? pk + P->params[o.lin_b].packed_off : pk + P->tproj_b_off);
            if (nt > 4)
                return launch_small_gemm(P->dtype, 0, (const float*)(ws + o.lin_in), o.lin_K, w, o.lin_K, nt, o.lin_N, o.lin_K, b, o.lin_silu,
                                         (float*)(ws + o.lin_out), o.lin_N, s, o.lin_pre >= 0 ? (float*)(ws + o.lin_pre) : nullptr);
            return launch_linear_wave(P->dtype, (const float*)(ws + o.lin_in), nt, o.lin_K, w, b, o.lin_N, o.lin_silu,
                                      (float*)(ws + o.lin_out), s);
        }
        case OP_GN:
            return run_gn(P, o, pk, ws, nt, drop_masks, s);
        case OP_CONV: {
            ConvArgs a{};
            fill_conv(P, o, pk, x, y, ws, drop_masks, nt, a, true);
            return run_any_conv(conv_dt(P, o), a, s);
        }
        case OP_CAST: {
            const Tensor& ts = P->tensors[o.cast_src];
            return launch_cast_f32_to_16(P->dtype, (const float*)(ws + ts.off), (int64_t)P->B * ts.H * ts.W * ts.C, ws + P->tensors[o.cast_dst].off, s);
        }
        case OP_ATTN: {
            const Tensor& q = P->tensors[o.at_qkv];
            const int S = q.H * q.W, C = q.C / 3;
            if (o.at_proj >= 0) {  // the block's proj conv + residual inside the launch (assign_attn_proj)
                ConvArgs a{};
                fill_conv(P, P->ops[o.at_proj], pk, x, y, ws, drop_masks, nt, a, true);
                return launch_attn_proj(P->dtype, ws + q.off, P->B, S, C, keep_ctx ? ws + P->tensors[o.at_out].off : nullptr, (float*)(ws + o.at_lse), a.w, a.bias,
                                        a.res1, a.dst, a.gn_part, a.gn_tiles, a.gn_cg, s);
            }
            if (P->x3 && attn_x3_supported(P->B, S, C, o.at_heads))  // (no log-sum-exp kept: the fp32 backward recomputes the scores)
                return launch_attn_x3(ws + q.off, P->B, S, C, o.at_heads, ws + P->tensors[o.at_out].off, s);
            if (o.at_heads > 1) {
                if (attn_heads_mfma_supported(P->dtype, P->B, S, C, o.at_heads))
                    return launch_attn_heads_mfma(P->dtype, ws + q.off, P->B, S, C, o.at_heads, ws + P->tensors[o.at_out].off, (float*)(ws + o.at_lse), s);
                return launch_attn_heads(P->dtype, ws + q.off, P->B, S, C, o.at_heads, ws + P->tensors[o.at_out].off, s);
            }
            if (attn_mfma_supported(P->dtype, P->B, S, C))
                return launch_attn_mfma(P->dtype, ws + q.off, P->B, S, C, ws + P->tensors[o.at_out].off, (float*)(ws + o.at_lse), s);
            return launch_attn_generic(P->dtype, ws + q.off, P->B, S, C, ws + P->tensors[o.at_out].off, s);
        }
    }
    return DMME_OK;
}
// kernel label + algorithmic flops / bytes of one op (bench.py's roofline accounting)
void op_account(const dmme_plan* P, const Op& o, char* label, int cap, double* flops, double* bytes) {
    const double es = (double)dtype_size(P->dtype);
    const char* tn = P->dtype == DMME_BF16 ? "bf16" : P->dtype == DMME_F16 ? "f16" : "float";  // (the conv labels add ":bf16x3" themselves)
    const double B = P->B;
    *flops = 0;
    *bytes = 0;
    if (o.lvl >= 0) {  // one launch for the whole run: accounted on its first op
        const LvlRun& R = P->lvl_runs[o.lvl];
        if (o.lvl_first) {
            snprintf(label, cap, "lvl_engine_kernel<%dx%d>", 1 << R.sh, 1 << R.sh);
            *flops = R.flops;
            *bytes = R.bytes;
        } else {
            snprintf(label, cap, "(level engine)");
        }
        return;
    }
    if (o.fused_away) {
        snprintf(label, cap, o.fused_away == 2 ? "(proj 1x1 conv inside its block's attention launch)" : "(residual 1x1 conv inside its block's conv2)");
        return;
    }
    switch (o.kind) {
        case OP_SINUS:
            snprintf(label, cap, "time_sinusoid_kernel");
            break;
        case OP_LINEAR:
            snprintf(label, cap, "linear_wave_kernel<%s>", tn);
            *flops = 2.0 * o.lin_K * o.lin_N;  // one timestep row when sampling
            *bytes = (double)o.lin_K * o.lin_N * es + 4.0 * (o.lin_K + 2.0 * o.lin_N);
            break;
        case OP_GN: {
            const Tensor& t1 = P->tensors[o.gn_src1];
            const int C = t1.C + (o.gn_src2 >= 0 ? P->tensors[o.gn_src2].C : 0);
            const bool fast = gn_fast_supported(P->dtype, P->B, t1.H * t1.W, t1.C, C - t1.C, P->cfg.num_groups);
            if (o.gn_direct || o.gn_in_consumer) {  // no launch: "(...)" labels are skipped by the per-kernel tables
                snprintf(label, cap, o.gn_direct ? "(norm finished by its producers' epilogues)" : "(norm finished by its consumer's parameter fill)");
            } else if (gn_from_parts(P, o)) {
                snprintf(label, cap, "gn_finalize_parts_kernel");
                *bytes = 2.0 * B * C * 4;
            } else {
                snprintf(label, cap, fast ? "gn_partial_kernel<%s>" : "gn_generic_kernel<%s>", tn);
                *bytes = B * t1.H * t1.W * C * es + 2.0 * B * C * 4;
            }
            break;
        }
        case OP_CONV: {
            ConvArgs a{};
            fill_conv(P, o, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a, true);
            const int cdt = conv_dt(P, o);
            if (conv_out_thin_supported(cdt, a))
                snprintf(label, cap, a.mix == 3 ? "conv_out_thin_kernel<%d,f16x3>" : "conv_out_thin_kernel<%d>", a.Cout * 9 <= 32 ? 1 : 2);
            else if (conv1x1_pipe_supported(cdt, a))
                conv1x1_pipe_label(cdt, a, label, cap);
            else if (conv_pipe_supported(cdt, a))
                conv_pipe_label(cdt, a, label, cap);
            else if (conv_mfma_supported(cdt, a))
                conv_mfma_label(cdt, a, label, cap);
            else
                snprintf(label, cap, "%s<%s>", conv_generic_kernel_name(a), o.route_f32 ? "float" : tn);
            const double Cin = a.C1 + a.C2, opix = B * a.Hout * a.Wout;
            *flops = 2.0 * opix * a.Cout * Cin * a.taps;
            *bytes = B * a.Hin * a.Win * Cin * (a.in_nchw ? 4.0 : es) + opix * a.Cout * (a.out_nchw ? 4.0 : es) +
                     (double)a.Cout * Cin * a.taps * es + (a.res1 ? opix * a.Cout * es : 0.0);
            if (a.r_w) {  // the residual segment: its raw input once, its filter, its products
                const double Cres = a.r_C1 + a.r_C2;
                *flops += 2.0 * opix * a.Cout * Cres;
                *bytes += opix * Cres * es + (double)a.Cout * Cres * es;
            }
            break;
        }
        case OP_CAST: {
            const Tensor& ts = P->tensors[o.cast_src];
            snprintf(label, cap, "cast_f32_to_16_kernel");
            *bytes = B * ts.H * ts.W * ts.C * 6.0;
            break;
        }
        case OP_ATTN: {
            const Tensor& q = P->tensors[o.at_qkv];
            const double S = q.H * q.W, C = q.C / 3;
            if (P->x3 && attn_x3_supported(P->B, (int)S, (int)C, o.at_heads))
                snprintf(label, cap, "attn_x3_kernel");
            else if (o.at_heads > 1)
                snprintf(label, cap, attn_heads_mfma_supported(P->dtype, P->B, (int)S, (int)C, o.at_heads) ? "attn_mfma_kernel<%s,heads>" : S == 16 ? "attn_s16_kernel<%s,heads>" : "attn_generic_kernel<%s,heads>", tn);
            else
                snprintf(label, cap, !attn_mfma_supported(P->dtype, P->B, (int)S, (int)C) ? (S == 16 ? "attn_s16_kernel<%s>" : "attn_generic_kernel<%s>") : attn_full_takes((int)S, (int)C) ? "attn_full_kernel<%s>" : "attn_mfma_kernel<%s>", tn);
            *flops = 4.0 * B * S * S * C;
            *bytes = B * S * 4.0 * C * es;
            if (o.at_proj >= 0) {  // + the proj conv and the residual: context neither written nor read, residual in, output out
                snprintf(label, cap, "attn_full_kernel<%s,proj>", tn);
                *flops += 2.0 * B * S * C * C;
                *bytes = B * S * 5.0 * C * es + C * C * es;
            }
            break;
        }
    }
}

}  // namespace dmme

// ======================================================================== extern "C"
extern "C" {

DMME_API const char* dmme_last_error(void) { return g_err; }
DMME_API int dmme_version(void) { return 105; }  // 105: dmme_attention_proj, dmme_unet_forward_nograd
DMME_API int dmme_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

DMME_API int dmme_unet_plan_create(const dmme_unet_cfg* cfg, int B, int H, int W, int dtype, int device, dmme_plan** out) {
    DMME_REQUIRE(cfg && out, DMME_ERR_INVALID, "plan_create: null argument");
    DMME_REQUIRE(B > 0 && H > 0 && W > 0, DMME_ERR_INVALID, "plan_create: bad shape B=%d H=%d W=%d", B, H, W);
    DMME_REQUIRE(dtype == DMME_F32 || dtype == DMME_BF16 || dtype == DMME_BF16X3 || dtype == DMME_F16 || dtype == DMME_F16R32, DMME_ERR_INVALID,
                 "plan_create: bad dtype %d", dtype);
    const int x3 = dtype == DMME_BF16X3;
    if (x3) dtype = DMME_F32;
    const int mix = dtype == DMME_F16R32;
    if (mix) {
        dtype = DMME_F16;  // (both architectures: the check below refuses a configuration whose fp32 level has a conv without a kernel)
    }
    DMME_REQUIRE(cfg->num_depths >= 1 && cfg->num_depths <= 8 && cfg->num_blocks >= 1, DMME_ERR_INVALID,
                 "plan_create: bad depth/blocks");
    DMME_REQUIRE(cfg->num_attention_depths >= 0 && cfg->num_attention_depths <= 8, DMME_ERR_INVALID, "bad attention_depths");
    DMME_REQUIRE(cfg->pos_dim >= 4 && cfg->pos_dim % 2 == 0, DMME_ERR_INVALID, "pos_dim must be even and >= 4");
    DMME_REQUIRE(cfg->arch == DMME_ARCH_DDPM || cfg->arch == DMME_ARCH_IDDPM, DMME_ERR_INVALID, "plan_create: unknown arch %d", cfg->arch);
    if (cfg->arch == DMME_ARCH_IDDPM) {
        DMME_REQUIRE(cfg->num_heads >= 1, DMME_ERR_INVALID, "plan_create: num_heads must be >= 1");
    }
    for (int d = 0; d < cfg->num_depths; ++d)
        DMME_REQUIRE(cfg->channels_per_depth[d] > 0 && cfg->channels_per_depth[d] % cfg->num_groups == 0,
                     DMME_ERR_INVALID, "channels_per_depth[%d]=%d not divisible by num_groups=%d", d,
                     cfg->channels_per_depth[d], cfg->num_groups);
    dmme_plan* P = new dmme_plan();
    P->cfg = *cfg;
    P->B = B;
    P->H = H;
    P->W = W;
    P->dtype = dtype;
    P->x3 = x3;
    P->mix = mix;
    P->device = device;
    int rc = build_plan(P);
    if (rc != DMME_OK) {
        delete P;
        return rc;
    }
    if (P->mix) {  // every conv of the fp32 level must have its kernel: there is no fall-back to a 16-bit path that would read fp32 as halves
        for (const Op& o : P->ops) {
            if (o.kind != OP_CONV || !(o.mix || o.route_f32)) continue;
            ConvArgs a{};
            fill_conv(P, o, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a);
            const bool ok = o.mix == 3 ? conv_out_thin_supported(P->dtype, a) : o.mix == 4 ? conv1x1_pipe_supported(P->dtype, a) : o.mix ? conv_pipe_supported(P->dtype, a) : true;
            if (!ok) {
                set_error("plan_create: precision fp16r32 has no kernel for the %dx%d conv %d+%d -> %d channels on the %dx%d level (B = %d)", o.taps == 9 ? 3 : 1,
                          o.taps == 9 ? 3 : 1, a.C1, a.C2, a.Cout, a.Hout, a.Wout, B);
                delete P;
                return DMME_ERR_UNSUPPORTED;
            }
        }
    }
    assign_levels(P);
    if (!getenv("DMME_NO_FUSED_GN")) assign_stats(P);
    if (!getenv("DMME_NO_FUSED_GN") && !getenv("DMME_NO_GN_DIRECT")) assign_direct(P);
    if (!getenv("DMME_NO_PREACT")) assign_preact(P);
    if (!getenv("DMME_NO_FUSED_GN")) assign_gn_in(P);
    assign_rseg(P);
    assign_attn_proj(P);
    assign_lvl_nograd(P);
    P->n_launches = 0;
    for (const Op& o : P->ops) {
        if (o.fused_away) continue;
        if (o.lvl >= 0)
            P->n_launches += o.lvl_first;
        else if (o.kind == OP_GN)
            P->n_launches += !(o.gn_direct || o.gn_in_consumer);
        else
            ++P->n_launches;
    }
    if (device >= 0) {
        // ---- gradient buckets: cuts at ResBlock starts (a block's first op is the GroupNorm of its conv1), walked in backward order
        std::vector<int> owner(P->params.size(), -1);  // op index that produces each parameter's gradient (-1: the time MLP, at the very end)
        for (int oi = 0; oi < (int)P->ops.size(); ++oi) {
            const Op& o = P->ops[oi];
            if (o.kind == OP_CONV) {
                owner[o.w] = oi;
                owner[o.b] = oi;
            } else if (o.kind == OP_GN) {
                owner[o.gn_gamma] = oi;
                owner[o.gn_beta] = oi;
            }
        }
        std::vector<int> tcol_owner(P->tblocks.size(), -1);
        for (size_t k = 0; k < P->tblocks.size(); ++k) {
            const auto& tb = P->tblocks[k];
            for (int oi = 0; oi < (int)P->ops.size(); ++oi) {
                const Op& o = P->ops[oi];
                if ((o.kind == OP_CONV && o.tproj_col == tb.col) || (o.kind == OP_GN && o.gn_mod_col == tb.col)) tcol_owner[k] = oi;
            }
            if (tcol_owner[k] >= 0) owner[tb.tw] = owner[tb.tb] = tcol_owner[k];
        }
        {
            int64_t total = 0;
            for (size_t pi = 0; pi < P->params.size(); ++pi)
                if (!P->params[pi].is_buffer) total += P->params[pi].numel();
            // block starts: a GroupNorm op whose consumer conv carries a time projection (DDPM) or that is followed by one (IDDPM conv1),
            // i.e. the first op of a ResBlock; also bare down / up convs.  Simpler and sufficient: any OP_GN whose source is not produced by
            // the op right before it inside the same block - approximated by "conv1's norm": the norm of a conv with tproj_col >= 0 (DDPM)
            // or the norm two ops ahead of a modulated norm (IDDPM).
            std::vector<char> is_start(P->ops.size(), 0);
            for (int oi = 0; oi < (int)P->ops.size(); ++oi) {
                const Op& o = P->ops[oi];
                if (o.kind != OP_CONV || o.gn < 0) continue;
                const bool conv1 = P->cfg.arch == DMME_ARCH_IDDPM ? (oi + 1 < (int)P->ops.size() && P->ops[oi + 1].kind == OP_GN && P->ops[oi + 1].gn_mod_col >= 0) : o.tproj_col >= 0;
                if (conv1) is_start[o.gn] = 1;
            }
            const int n_target = debug_route("grad_buckets", 6);
            // candidates: block starts with the fraction of the parameters backward has finished when the walk reaches them
            std::vector<std::pair<int, double>> cand;
            {
                int64_t acc = 0;
                for (int oi = (int)P->ops.size() - 1; oi > 0; --oi) {
                    for (size_t pi = 0; pi < P->params.size(); ++pi)
                        if (owner[pi] == oi && !P->params[pi].is_buffer) acc += P->params[pi].numel();
                    if (is_start[oi]) cand.push_back({oi, (double)acc / (double)(total > 0 ? total : 1)});
                }
            }
            std::vector<int> cuts{(int)P->ops.size()};
            if (n_target > 1 && !cand.empty()) {
                // the last cut first: what is left behind it (first down blocks, input conv, time MLP) is the one exchange no compute
                // hides - as close to 12 % of the bytes as the block boundaries allow
                int last = -1;
                double best = 1e9;
                for (int k = 0; k < (int)cand.size(); ++k) {
                    const double rest = 1.0 - cand[k].second;
                    if (rest < 0.04) continue;
                    const double d = rest > 0.12 ? rest - 0.12 : 2.0 * (0.12 - rest);
                    if (d < best) { best = d; last = k; }
                }
                if (last >= 0) {
                    // the others: the block boundary nearest to each multiple of (what is in front of the last cut) / (n - 1)
                    const double step = cand[last].second / (double)(n_target - 1);
                    int prev_k = -1;
                    for (int q = 1; q < n_target - 1; ++q) {
                        int pick = -1;
                        double bd = 1e9;
                        for (int k = prev_k + 1; k < last; ++k) {
                            const double d = cand[k].second > q * step ? cand[k].second - q * step : q * step - cand[k].second;
                            if (d < bd) { bd = d; pick = k; }
                        }
                        if (pick < 0) break;
                        cuts.push_back(cand[pick].first);
                        prev_k = pick;
                    }
                    cuts.push_back(cand[last].first);
                }
            }
            cuts.push_back(0);
            bool clean = cuts.size() > 2 && !getenv("DMME_NO_GRAD_BUCKETS");
            if (clean) {
                P->gb.resize(cuts.size() - 1);
                for (size_t b = 0; b + 1 < cuts.size(); ++b) {
                    dmme_plan::GradBucket& G = P->gb[b];
                    G.op_hi = cuts[b];
                    G.op_lo = cuts[b + 1];
                    const bool last = b + 2 == cuts.size();
                    std::vector<std::pair<int64_t, int64_t>> r;
                    for (size_t pi = 0; pi < P->params.size(); ++pi) {
                        const Param& pp = P->params[pi];  // (the sinusoid table, a buffer without gradient, rides in the last bucket: the
                                                          // hand-overs then tile the whole flat buffer)
                        const bool mine = owner[pi] < 0 ? last : (owner[pi] >= G.op_lo && owner[pi] < G.op_hi);
                        if (!mine) continue;
                        if (!r.empty() && r.back().first + r.back().second == pp.ref_off) r.back().second += pp.numel();
                        else r.push_back({pp.ref_off, pp.numel()});
                    }
                    G.ranges = r;
                    for (size_t k = 0; k < P->tblocks.size(); ++k) {
                        if (tcol_owner[k] < G.op_lo || tcol_owner[k] >= G.op_hi) continue;
                        const int c0 = P->tblocks[k].col, c1 = c0 + P->tblocks[k].cout;
                        if (!G.tcols.empty() && G.tcols.back().second == c0) G.tcols.back().second = c1;
                        else G.tcols.push_back({c0, c1});
                    }
                    // (the tiled time-projection gradient addresses 64-column tiles: it exists only when every block's width is a
                    // multiple of 64, and then so is every range start)
                }
                for (size_t k = 0; k < P->tblocks.size(); ++k)
                    if (tcol_owner[k] < 0) clean = false;
            }
            if (!clean) P->gb.clear();
        }
        for (auto& G : P->gb)  // before the "all" build: that one leaves the final Op::wg_layer values
            for (int k = 0; k < 3; ++k) build_wgrad_group(P, G.wg[k], k, G.op_lo, G.op_hi);
        for (int k = 0; k < 3; ++k) build_wgrad_group(P, P->wg[k], k);
        std::vector<int> bias_job_op, col_job_op;  // op index each job belongs to (gradient buckets)
        if (!debug_route("no_bias_group"))
            for (Op& o : P->ops) {
                if (o.kind != OP_CONV) continue;
                const int o_index = (int)(&o - P->ops.data());
                ConvArgs a{};
                fill_conv(P, o, nullptr, nullptr, nullptr, nullptr, nullptr, 1, a);
                if (o.gn >= 0 && o.b_gnrows >= 0) {
                    // the norm in front of this conv: its backward leaves per-image dbeta / dgamma rows, summed over the batch by the same
                    // grouped launch as the biases.  Bucket (gradient exchange overlap): by the NORM's op index, its jobs first.
                    const Op& gop = P->ops[o.gn];
                    const Tensor& t1 = P->tensors[gop.gn_src1];
                    const int C1 = t1.C, C2 = gop.gn_src2 >= 0 ? P->tensors[gop.gn_src2].C : 0, C = C1 + C2;
                    if (gn_bwd_fast_supported(P->dtype, t1.H * t1.W, C1, C2) &&
                        gn_bwd_rows_supported(P->dtype, t1.H * t1.W, C1, C2, P->cfg.num_groups, gop.gn_mod_col >= 0)) {
                        o.gn_rows_deferred = 1;
                        for (int which = 0; which < 2; ++which)
                            for (int cb = 0; cb < (C + 31) / 32; ++cb) {
                                BiasJob j{};
                                j.rowsum_off = o.b_gnrows + (int64_t)which * P->B * C * 4;
                                j.dbias_off = P->params[which == 0 ? gop.gn_beta : gop.gn_gamma].ref_off;
                                j.C = C;
                                j.cblock = cb;
                                j.tcol = -1;
                                P->bias_jobs.push_back(j);
                                bias_job_op.push_back(o.gn);
                            }
                    }
                }
                if (!colsum_fast_supported(P->dtype, a.Hout * a.Wout, a.Cout)) continue;
                o.bias_deferred = 1;
                if (!debug_route("no_colsum_group")) {
                    ColJob cj{};
                    const int nch = colsum_group_chunks(P->dtype, a.Hout * a.Wout, a.Cout, &cj.chunk_px, &cj.ppw);
                    cj.dy_off = o.dst == -2 ? P->bws_dy : P->gt_off[o.dst];
                    cj.rowsum_off = o.b_rowsum;
                    cj.HW = a.Hout * a.Wout;
                    cj.C = a.Cout;
                    for (int ch = 0; ch < nch; ++ch) {
                        cj.chunk = ch;
                        P->col_jobs.push_back(cj);
                        col_job_op.push_back(o_index);
                    }
                }
                for (int cb = 0; cb < (a.Cout + 31) / 32; ++cb) {
                    BiasJob j{};
                    j.rowsum_off = o.b_rowsum;
                    j.dbias_off = P->params[o.b].ref_off;
                    j.C = a.Cout;
                    j.cblock = cb;
                    j.tcol = o.tproj_col;
                    P->bias_jobs.push_back(j);
                    bias_job_op.push_back(o_index);
                }
            }
        for (auto& G : P->gb) {  // jobs were pushed in ascending op order: a bucket's jobs are one index range
            auto range = [&](const std::vector<int>& ops_of, int& j0, int& j1) {
                j0 = j1 = 0;
                bool any = false, ok = true;
                for (int j = 0; j < (int)ops_of.size(); ++j) {
                    if (ops_of[j] < G.op_lo || ops_of[j] >= G.op_hi) continue;
                    if (!any) { j0 = j; any = true; } else if (j != j1) ok = false;
                    j1 = j + 1;
                }
                return ok;
            };
            if (!range(bias_job_op, G.bias0, G.bias1) || !range(col_job_op, G.col0, G.col1)) {
                P->gb.clear();
                break;
            }
        }
    }
    if (device >= 0) {
        std::vector<PackItem> items;
        build_pack_items(P, items);
        P->n_items = (int)items.size();
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc((void**)&P->items_dev, items.size() * sizeof(PackItem));
        if (e == hipSuccess) e = hipMemcpy(P->items_dev, items.data(), items.size() * sizeof(PackItem), hipMemcpyHostToDevice);
        std::vector<PackItem> bitems;
        build_pack_items_bwd(P, bitems);
        P->n_items_bwd = (int)bitems.size();
        if (e == hipSuccess) e = hipMalloc((void**)&P->items_bwd_dev, bitems.size() * sizeof(PackItem));
        if (e == hipSuccess) e = hipMemcpy(P->items_bwd_dev, bitems.data(), bitems.size() * sizeof(PackItem), hipMemcpyHostToDevice);
        std::vector<PackItem> uitems;
        build_unpack_items(P, uitems);
        P->n_items_unpack = (int)uitems.size();
        if (e == hipSuccess) e = hipMalloc((void**)&P->items_unpack_dev, uitems.size() * sizeof(PackItem));
        if (e == hipSuccess) e = hipMemcpy(P->items_unpack_dev, uitems.data(), uitems.size() * sizeof(PackItem), hipMemcpyHostToDevice);
        {
            bool ok = !P->tblocks.empty() && P->tproj_cols % 64 == 0;
            for (const auto& tb : P->tblocks) ok = ok && tb.cout % 64 == 0 && tb.col % 64 == 0;
            if (ok) {
                std::vector<int64_t> tiles(P->tproj_cols / 64 + P->tproj_cols / 32, -1);
                const int n64 = P->tproj_cols / 64;
                for (const auto& tb : P->tblocks) {
                    for (int r = 0; r < tb.cout; r += 64) tiles[(tb.col + r) / 64] = P->params[tb.tw].ref_off + (int64_t)r * P->cfg.emb_dim;
                    for (int r = 0; r < tb.cout; r += 32) tiles[n64 + (tb.col + r) / 32] = P->params[tb.tb].ref_off + r;
                }
                for (int64_t v : tiles) ok = ok && v >= 0;
                if (ok) {
                    if (e == hipSuccess) e = hipMalloc((void**)&P->tp_tiles_dev, tiles.size() * sizeof(int64_t));
                    if (e == hipSuccess) e = hipMemcpy(P->tp_tiles_dev, tiles.data(), tiles.size() * sizeof(int64_t), hipMemcpyHostToDevice);
                    P->tp_n64 = n64;
                }
            }
        }
        if (!P->col_jobs.empty()) {
            if (e == hipSuccess) e = hipMalloc((void**)&P->col_jobs_dev, P->col_jobs.size() * sizeof(ColJob));
            if (e == hipSuccess) e = hipMemcpy(P->col_jobs_dev, P->col_jobs.data(), P->col_jobs.size() * sizeof(ColJob), hipMemcpyHostToDevice);
        }
        if (!P->bias_jobs.empty()) {
            if (e == hipSuccess) e = hipMalloc((void**)&P->bias_jobs_dev, P->bias_jobs.size() * sizeof(BiasJob));
            if (e == hipSuccess) e = hipMemcpy(P->bias_jobs_dev, P->bias_jobs.data(), P->bias_jobs.size() * sizeof(BiasJob), hipMemcpyHostToDevice);
        }
        std::vector<dmme_plan::WgGroup*> all_groups{&P->wg[0], &P->wg[1], &P->wg[2]};
        for (auto& B_ : P->gb)
            for (int k = 0; k < 3; ++k) all_groups.push_back(&B_.wg[k]);
        for (dmme_plan::WgGroup* G : all_groups) {
            if (G->jobs.empty()) continue;
            if (e == hipSuccess) e = hipMalloc((void**)&G->layers_dev, G->layers.size() * sizeof(WgLayer));
            if (e == hipSuccess) e = hipMemcpy(G->layers_dev, G->layers.data(), G->layers.size() * sizeof(WgLayer), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void**)&G->jobs_dev, G->jobs.size() * sizeof(WgJob));
            if (e == hipSuccess) e = hipMemcpy(G->jobs_dev, G->jobs.data(), G->jobs.size() * sizeof(WgJob), hipMemcpyHostToDevice);
        }
        if (!P->lvl_runs.empty()) {
            if (e == hipSuccess) e = hipHostMalloc((void**)&P->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent);
            if (e == hipSuccess) memset(P->err_host, 0, 64);
        }
        for (LvlRun& R : P->lvl_runs) {
            const size_t words = 16 + R.ops.size() * 2 * (size_t)R.NG * LVL_NS;
            if (e == hipSuccess) e = hipMalloc((void**)&R.ops_dev, R.ops.size() * sizeof(LvlOp));
            if (e == hipSuccess) e = hipMemcpy(R.ops_dev, R.ops.data(), R.ops.size() * sizeof(LvlOp), hipMemcpyHostToDevice);
            if (e == hipSuccess && R.raw_skipped > 0) {
                e = hipMalloc((void**)&R.ops_nograd_dev, R.ops_nograd.size() * sizeof(LvlOp));
                if (e == hipSuccess) e = hipMemcpy(R.ops_nograd_dev, R.ops_nograd.data(), R.ops_nograd.size() * sizeof(LvlOp), hipMemcpyHostToDevice);
            }
            if (e == hipSuccess) e = hipMalloc((void**)&R.sync_dev, words * 4);
            if (e == hipSuccess) e = hipMemset(R.sync_dev, 0, words * 4);
        }
        for (auto& G : P->gb)  // unpack items follow the parameter order: a bucket's items are the runs inside its flat ranges
            for (int i = 0; i < (int)uitems.size(); ++i) {
                bool mine = false;
                for (auto& r : G.ranges) mine = mine || (uitems[i].src_off >= r.first && uitems[i].src_off < r.first + r.second);
                if (!mine) continue;
                if (!G.unpack.empty() && G.unpack.back().second == i) G.unpack.back().second = i + 1;
                else G.unpack.push_back({i, i + 1});
            }
        if (e != hipSuccess) {
            set_error("plan_create: device table setup failed: %s", hipGetErrorString(e));
            delete P;
            return DMME_ERR_HIP;
        }
    }
    *out = P;
    return DMME_OK;
}

DMME_API void dmme_unet_plan_destroy(dmme_plan* plan) {
    if (!plan) return;
    if (plan->items_dev) (void)hipFree(plan->items_dev);
    if (plan->items_bwd_dev) (void)hipFree(plan->items_bwd_dev);
    if (plan->items_unpack_dev) (void)hipFree(plan->items_unpack_dev);
    std::vector<dmme_plan::WgGroup*> all_groups{&plan->wg[0], &plan->wg[1], &plan->wg[2]};
    for (auto& B_ : plan->gb)
        for (int k = 0; k < 3; ++k) all_groups.push_back(&B_.wg[k]);
    for (dmme_plan::WgGroup* G : all_groups) {
        if (G->layers_dev) (void)hipFree(G->layers_dev);
        if (G->jobs_dev) (void)hipFree(G->jobs_dev);
    }
    if (plan->tp_tiles_dev) (void)hipFree(plan->tp_tiles_dev);
    if (plan->bias_jobs_dev) (void)hipFree(plan->bias_jobs_dev);
    if (plan->col_jobs_dev) (void)hipFree(plan->col_jobs_dev);
    if (plan->err_host) (void)hipHostFree(plan->err_host);
    for (LvlRun& R : plan->lvl_runs) {
        if (R.ops_dev) (void)hipFree(R.ops_dev);
        if (R.ops_nograd_dev) (void)hipFree(R.ops_nograd_dev);
        if (R.sync_dev) (void)hipFree(R.sync_dev);
    }
    delete plan;
}

DMME_API int dmme_unet_plan_num_params(const dmme_plan* plan) { return plan ? (int)plan->params.size() : 0; }

DMME_API int dmme_unet_plan_param_info(const dmme_plan* plan, int index, char* name, int name_cap, int* ndim,
                              int64_t shape[4], int64_t* ref_offset, int* is_buffer) {
    DMME_REQUIRE(plan && index >= 0 && index < (int)plan->params.size(), DMME_ERR_INVALID, "param_info: bad index %d", index);
    const Param& p = plan->params[index];
    if (name && name_cap > 0) {
        strncpy(name, p.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (ndim) *ndim = p.ndim;
    if (shape)
        for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
    if (ref_offset) *ref_offset = p.ref_off;
    if (is_buffer) *is_buffer = p.is_buffer ? 1 : 0;
    return DMME_OK;
}

DMME_API int64_t dmme_unet_plan_ref_numel(const dmme_plan* plan) { return plan ? plan->ref_numel : 0; }
DMME_API int64_t dmme_unet_plan_packed_bytes(const dmme_plan* plan) { return plan ? plan->packed_bytes : 0; }
DMME_API int64_t dmme_unet_plan_workspace_bytes(const dmme_plan* plan) { return plan ? plan->ws_bytes : 0; }
DMME_API int64_t dmme_unet_plan_dropmask_numel(const dmme_plan* plan) { return plan ? plan->dropmask_numel : 0; }
DMME_API int dmme_unet_plan_out_channels(const dmme_plan* plan) { return plan ? plan->out_channels : 0; }
DMME_API int dmme_unet_plan_num_launches(const dmme_plan* plan) { return plan ? plan->n_launches : 0; }

DMME_API int dmme_unet_pack_params(const dmme_plan* plan, const float* ref_flat, void* packed, void* stream) {
    DMME_REQUIRE(plan && ref_flat && packed, DMME_ERR_INVALID, "pack_params: null argument");
    DMME_REQUIRE(plan->items_dev, DMME_ERR_INVALID, "pack_params: plan was created without a device");
    return launch_pack_table(plan->dtype, plan->items_dev, plan->n_items, ref_flat, packed, (hipStream_t)stream);
}

static int unet_forward_impl(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t, int t_len,
                      float* y, void* workspace, const float* drop_masks, void* stream, bool keep_ctx) {
    DMME_REQUIRE(plan && packed && x && t && y && workspace, DMME_ERR_INVALID, "unet_forward: null argument");
    DMME_REQUIRE(t_len == 1 || t_len == plan->B, DMME_ERR_INVALID,
                 "unet_forward: timestep tensor of length %d does not broadcast against batch %d", t_len, plan->B);
    {
        const int rc = lvl_check(plan, "unet_forward", (hipStream_t)stream, true);
        if (rc != DMME_OK) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    const dmme_plan* P = plan;
    const char* pk = (const char*)packed;
    char* ws = (char*)workspace;
    const int nt = t_len;
    if (keep_ctx) {
        if (plan->nograd_ws == workspace) plan->nograd_ws = nullptr;
    } else {
        plan->nograd_ws = workspace;
    }
    for (const Op& o : P->ops) {
        const int rc = run_op(P, o, pk, x, t, nt, y, ws, drop_masks, s, keep_ctx);
        if (rc != DMME_OK) return rc;
    }
    return DMME_OK;
}
DMME_API int dmme_unet_forward(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t, int t_len,
                      float* y, void* workspace, const float* drop_masks, void* stream) {
    return unet_forward_impl(plan, packed, x, t, t_len, y, workspace, drop_masks, stream, true);
}
// the same forward where no backward pass will follow (sampling, evaluation under no_grad): tensors only a backward pass reads - the
// context of an attention block whose proj conv runs inside the attention launch - are not written
DMME_API int dmme_unet_forward_nograd(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t, int t_len,
                      float* y, void* workspace, const float* drop_masks, void* stream) {
    return unet_forward_impl(plan, packed, x, t, t_len, y, workspace, drop_masks, stream, false);
}

DMME_API int dmme_unet_plan_num_ops(const dmme_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

DMME_API int dmme_unet_plan_op_info(const dmme_plan* plan, int index, char* label, int label_cap, double* flops,
                                    double* bytes) {
    DMME_REQUIRE(plan && index >= 0 && index < (int)plan->ops.size() && label && label_cap > 0 && flops && bytes,
                 DMME_ERR_INVALID, "op_info: bad argument");
    op_account(plan, plan->ops[index], label, label_cap, flops, bytes);
    return DMME_OK;
}

DMME_API int dmme_unet_forward_profiled(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t,
                                        int t_len, float* y, void* workspace, const float* drop_masks, void* stream,
                                        float* op_ms) {
    DMME_REQUIRE(plan && packed && x && t && y && workspace && op_ms, DMME_ERR_INVALID, "forward_profiled: null argument");
    DMME_REQUIRE(t_len == 1 || t_len == plan->B, DMME_ERR_INVALID, "forward_profiled: bad t_len %d", t_len);
    if (int rc0 = lvl_check(plan, "forward_profiled", (hipStream_t)stream, true)) return rc0;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = plan->ops.size();
    plan->nograd_ws = workspace;
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) DMME_CHECK_HIP(hipEventCreate(&e));
    int rc = DMME_OK;
    DMME_CHECK_HIP(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < n && rc == DMME_OK; ++i) {
        rc = run_op(plan, plan->ops[i], (const char*)packed, x, t, t_len, y, (char*)workspace, drop_masks, s, false);  // (the sampling step's forward)
        if (rc == DMME_OK && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = DMME_ERR_HIP;
    }
    if (rc == DMME_OK && hipEventSynchronize(ev[n]) != hipSuccess) rc = DMME_ERR_HIP;
    for (size_t i = 0; i < n && rc == DMME_OK; ++i)
        if (hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = DMME_ERR_HIP;
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc == DMME_ERR_HIP) set_error("forward_profiled: HIP event error");
    return rc;
}

// ---------------------------------------------------------------- training: backward + optimiser
DMME_API int dmme_grad_norm(const float* grad, int64_t numel, float* norm_out, float* scratch, void* stream) {
    DMME_REQUIRE(grad && norm_out && scratch && numel > 0, DMME_ERR_INVALID, "grad_norm: bad argument");
    return launch_grad_norm(grad, numel, norm_out, scratch, (hipStream_t)stream);
}

DMME_API int dmme_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr,
                            float beta1, float beta2, float eps, int step, const float* grad_norm, float max_norm, float ema_decay,
                            float grad_scale, void* stream) {
    DMME_REQUIRE(param && grad && exp_avg && exp_avg_sq && numel > 0 && step >= 1 && grad_scale > 0.f, DMME_ERR_INVALID, "adam_step: bad argument");
    return launch_adam(param, grad, exp_avg, exp_avg_sq, ema, numel, lr, beta1, beta2, eps, step, grad_norm, max_norm, ema_decay, grad_scale,
                       (hipStream_t)stream);
}

DMME_API int dmme_amp_init(float* amp_state, float init_scale, void* stream) {
    DMME_REQUIRE(amp_state && init_scale > 0.f, DMME_ERR_INVALID, "amp_init: bad argument");
    return launch_amp_init(amp_state, init_scale, (hipStream_t)stream);
}
DMME_API int dmme_amp_scale(float* grad, int64_t numel, const float* amp_state, void* stream) {
    DMME_REQUIRE(grad && amp_state && numel >= 0, DMME_ERR_INVALID, "amp_scale: bad argument");
    return launch_amp_scale(grad, numel, amp_state, (hipStream_t)stream);
}
DMME_API int dmme_adam_step_amp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr, float beta1,
                                float beta2, float eps, const float* grad_norm, float max_grad_norm, float ema_decay, float grad_scale,
                                float* amp_state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    DMME_REQUIRE(param && grad && exp_avg && exp_avg_sq && grad_norm && amp_state && numel >= 0, DMME_ERR_INVALID, "adam_step_amp: null argument");
    DMME_REQUIRE(growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1, DMME_ERR_INVALID, "adam_step_amp: bad scaler constants");
    return launch_adam_amp(param, grad, exp_avg, exp_avg_sq, ema, numel, lr, beta1, beta2, eps, grad_norm, max_grad_norm, ema_decay, grad_scale, amp_state,
                           growth_factor, backoff_factor, growth_interval, (hipStream_t)stream);
}

DMME_API int dmme_grad_pack_bf16(const float* grad, int64_t numel, void* dst_bf16, int64_t numel_padded, void* stream) {
    DMME_REQUIRE(grad && dst_bf16 && numel > 0 && numel_padded >= numel, DMME_ERR_INVALID, "grad_pack_bf16: bad argument");
    return launch_grad_pack_bf16(grad, numel, dst_bf16, numel_padded, (hipStream_t)stream);
}
DMME_API int dmme_shard_reduce_bf16(const void* recv_bf16, int world, int64_t per_rank, float scale, void* out_bf16, void* stream) {
    DMME_REQUIRE(recv_bf16 && out_bf16 && world >= 1 && per_rank > 0, DMME_ERR_INVALID, "shard_reduce_bf16: bad argument");
    return launch_shard_reduce_bf16(recv_bf16, world, per_rank, scale, out_bf16, (hipStream_t)stream);
}
DMME_API int dmme_grad_unpack_bf16(const void* src_bf16, int64_t numel, float* grad, void* stream) {
    DMME_REQUIRE(src_bf16 && grad && numel > 0, DMME_ERR_INVALID, "grad_unpack_bf16: bad argument");
    return launch_grad_unpack_bf16(src_bf16, numel, grad, (hipStream_t)stream);
}

DMME_API int dmme_unet_debug_read(const dmme_plan* plan, const void* workspace, const char* name, float* dst,
                         int64_t numel_cap, int64_t* numel_out, void* stream) {
    DMME_REQUIRE(plan && workspace && name && dst, DMME_ERR_INVALID, "debug_read: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (strcmp(name, "condition") == 0) {
        const int64_t n = (int64_t)plan->B * plan->cfg.emb_dim;  // rows beyond t_len are unspecified
        DMME_REQUIRE(n <= numel_cap, DMME_ERR_INVALID, "debug_read: destination too small");
        DMME_CHECK_HIP(hipMemcpyAsync(dst, (const char*)workspace + plan->ws_temb, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (numel_out) *numel_out = n;
        return DMME_OK;
    }
    auto it = plan->named.find(name);
    DMME_REQUIRE(it != plan->named.end(), DMME_ERR_INVALID, "debug_read: unknown module '%s'", name);
    const Tensor& t = plan->tensors[it->second];
    const int64_t n = (int64_t)plan->B * t.C * t.H * t.W;
    DMME_REQUIRE(n <= numel_cap, DMME_ERR_INVALID, "debug_read: destination too small (%lld > %lld)", (long long)n,
                 (long long)numel_cap);
    if (numel_out) *numel_out = n;
    return launch_nhwc_to_nchw(t.f32 ? DMME_F32 : plan->dtype, (const char*)workspace + t.off, plan->B, t.C, t.H * t.W, dst, s);
}

DMME_API int dmme_dropout_masks(const dmme_plan* plan, uint64_t seed, uint64_t offset, float* masks, void* stream) {
    DMME_REQUIRE(plan && masks, DMME_ERR_INVALID, "dropout_masks: null argument");
    return launch_dropmask(masks, plan->dropmask_numel, plan->cfg.dropout, seed, offset, (hipStream_t)stream);
}

DMME_API int dmme_randn(float* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream) {
    DMME_REQUIRE(out && numel >= 0, DMME_ERR_INVALID, "randn: bad argument");
    return launch_randn(out, numel, seed, offset, (hipStream_t)stream);
}

DMME_API int dmme_q_sample(const float* x0, const float* z, const float* sqrt_abar, const float* sqrt_1m_abar,
                           const int64_t* t, int B, int64_t chw, float* x_t, float* target, void* stream) {
    DMME_REQUIRE(x0 && z && sqrt_abar && sqrt_1m_abar && t && x_t && B > 0 && chw > 0, DMME_ERR_INVALID, "q_sample: bad argument");
    return launch_q_sample(x0, z, sqrt_abar, sqrt_1m_abar, t, B, chw, x_t, target, (hipStream_t)stream);
}

DMME_API int dmme_ddpm_step(float* x, const float* eps, const float* z, float inv_sqrt_alpha, float eps_coef, float sigma,
                   int add_noise, int64_t numel, void* stream) {
    DMME_REQUIRE(x && eps && (z || !add_noise), DMME_ERR_INVALID, "ddpm_step: null argument");
    return launch_ddpm_step(x, eps, z, inv_sqrt_alpha, eps_coef, sigma, add_noise, numel, (hipStream_t)stream);
}

DMME_API int dmme_ddim_step(float* x, const float* eps, float sqrt_one_minus_abar, float sqrt_abar_prev, int64_t numel,
                   void* stream) {
    DMME_REQUIRE(x && eps, DMME_ERR_INVALID, "ddim_step: null argument");
    return launch_ddim_step(x, eps, sqrt_one_minus_abar, sqrt_abar_prev, numel, (hipStream_t)stream);
}

DMME_API int dmme_chain_set(void* state, int64_t i, const int64_t* t_table, uint64_t philox_seed, uint64_t philox_offset, void* stream) {
    DMME_REQUIRE(state && t_table && i >= 0, DMME_ERR_INVALID, "chain_set: bad argument");
    return launch_chain_set(state, i, t_table, philox_seed, philox_offset, (hipStream_t)stream);
}

DMME_API int dmme_chain_update(int kind, float* x, const float* model_out, const float* step_coef, const int64_t* t_table, void* state,
                               int B, int64_t chw, void* stream) {
    DMME_REQUIRE(x && model_out && step_coef && t_table && state && B > 0 && chw > 0, DMME_ERR_INVALID, "chain_update: bad argument");
    return launch_chain_update(kind, x, model_out, step_coef, t_table, state, B, chw, (hipStream_t)stream);
}

DMME_API int dmme_chain_step(const dmme_plan* plan, const void* packed, float* x, float* model_out, void* workspace, int kind,
                             const float* step_coef, const int64_t* t_table, void* state, void* stream) {
    DMME_REQUIRE(plan && packed && x && model_out && workspace && step_coef && t_table && state, DMME_ERR_INVALID, "chain_step: null argument");
    if (int rc0 = lvl_check(plan, "chain_step", (hipStream_t)stream, true)) return rc0;
    DMME_REQUIRE((kind == DMME_CHAIN_IDDPM) == (plan->out_channels == 2 * plan->cfg.in_channels), DMME_ERR_INVALID,
                 "chain_step: sampler kind %d does not fit a network with %d output channels", kind, plan->out_channels);
    // the timestep the network is evaluated at is the second word of the device-resident loop state
    const int64_t* t_dev = (const int64_t*)state + 1;
    int rc = unet_forward_impl(plan, packed, x, t_dev, 1, model_out, workspace, nullptr, stream, false);
    if (rc != DMME_OK) return rc;
    return launch_chain_update(kind, x, model_out, step_coef, t_table, state, plan->B, (int64_t)plan->cfg.in_channels * plan->H * plan->W,
                               (hipStream_t)stream);
}

DMME_API int dmme_image_batch(const uint8_t* data, int64_t n_images, const int64_t* idx, const uint8_t* flip, int B, int C, int H, int W,
                              float* out, void* stream) {
    DMME_REQUIRE(data && idx && out && n_images > 0 && B > 0 && C > 0 && H > 0 && W > 0, DMME_ERR_INVALID, "image_batch: bad argument");
    return launch_image_batch(data, idx, flip, B, C, H, W, out, (hipStream_t)stream);
}

DMME_API int dmme_iddpm_step(float* x, const float* model_out, const float* z, float inv_sqrt_alpha, float eps_coef, float log_beta,
                             float log_beta_tilde, int add_noise, int B, int64_t chw, void* stream) {
    DMME_REQUIRE(x && model_out && (z || !add_noise) && B > 0 && chw > 0, DMME_ERR_INVALID, "iddpm_step: bad argument");
    return launch_iddpm_step(x, model_out, z, inv_sqrt_alpha, eps_coef, log_beta, log_beta_tilde, add_noise, B, chw, (hipStream_t)stream);
}

DMME_API int dmme_iddpm_loss(const float* model_out, const float* x_t, const float* x_0, const float* target, const int64_t* t,
                             const float* coef, int B, int64_t chw, float w_simple, float w_vlb, float* loss, float* d_out,
                             float grad_scale, float* scratch, void* stream) {
    DMME_REQUIRE(model_out && x_t && x_0 && target && t && coef && loss && scratch && B > 0 && chw > 0, DMME_ERR_INVALID,
                 "iddpm_loss: bad argument");
    return launch_iddpm_loss(model_out, x_t, x_0, target, t, coef, B, chw, w_simple, w_vlb, loss, d_out, grad_scale, scratch,
                             (hipStream_t)stream);
}

DMME_API int dmme_mse_loss(const float* eps, const float* target, int64_t numel, float* loss, float* d_eps, float grad_scale,
                  float* scratch, void* stream) {
    DMME_REQUIRE(eps && target && loss && scratch, DMME_ERR_INVALID, "mse_loss: null argument");
    return launch_mse(eps, target, numel, loss, d_eps, grad_scale, scratch, (hipStream_t)stream);
}

DMME_API int dmme_debug_issue_probe(int kind, int n_inner, int iters, int flags, int blocks, void* sink, const void* src, void* stream) {
    return launch_issue_probe(kind, n_inner, iters, flags, blocks, (long long*)sink, src, (hipStream_t)stream);
}
DMME_API int dmme_debug_mfma_valu(int mode, int iters, int blocks, void* sink, void* stream) {
    DMME_REQUIRE(sink && iters > 0 && blocks > 0, DMME_ERR_INVALID, "mfma_valu: bad argument");
    return launch_mfma_valu(mode, iters, blocks, (float*)sink, (hipStream_t)stream);
}
DMME_API int dmme_debug_l2_stream(const void* buf, int64_t bytes, int iters, int mode, int depth, int blocks, void* sink, void* stream) {
    return launch_l2_stream(buf, bytes, iters, mode, depth, blocks, (unsigned*)sink, (hipStream_t)stream);
}

static long long* g_stamps = nullptr;
DMME_API int dmme_debug_set_stamps(void* buf) {
    g_stamps = (long long*)buf;
    return DMME_OK;
}

DMME_API int dmme_conv2d(const dmme_conv_desc* d, const void* src1, const void* src2, const void* weight, const float* bias,
                const float* scale, const float* shift, const float* dmask, const float* tproj, const void* res1,
                const void* res2, int R1, void* dst, void* stream) {
    DMME_REQUIRE(d && src1 && weight && bias && dst, DMME_ERR_INVALID, "conv2d: null argument");
    DMME_REQUIRE(d->taps == 9 || d->taps == 1, DMME_ERR_INVALID, "conv2d: taps must be 1 or 9");
    DMME_REQUIRE(d->stride == 1 || d->stride == 2, DMME_ERR_INVALID, "conv2d: stride must be 1 or 2");
    ConvArgs a{};
    a.src1 = src1; a.src2 = src2; a.w = weight; a.bias = bias; a.scale = scale; a.shift = shift; a.dmask = dmask;
    a.tproj = d->nt > 0 ? tproj : nullptr;
    a.res1 = res1; a.res2 = res2; a.R1 = R1; a.dst = dst;
    a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = src2 ? d->C2 : 0;
    a.up = d->upsample; a.stride = d->stride; a.taps = d->taps;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    a.Hout = Hv / a.stride; a.Wout = Wv / a.stride; a.Cout = d->Cout;
    a.pro_silu = d->pro_silu; a.out_silu = d->out_silu; a.nt = d->nt; a.tproj_ld = d->tproj_ld;
    a.in_nchw = d->in_nchw; a.out_nchw = d->out_nchw;
    a.stamps = g_stamps;
    DMME_REQUIRE(d->dtype == DMME_F32 || d->dtype == DMME_BF16 || d->dtype == DMME_BF16X3 || d->dtype == DMME_F16 || d->dtype == DMME_F16R32, DMME_ERR_INVALID,
                 "conv2d: bad dtype %d", d->dtype);
    a.x3 = d->dtype == DMME_BF16X3;
    a.f16 = d->dtype == DMME_F16;
    if (d->dtype == DMME_F16R32) {
        // the split-pass kernels of the mixed mode's fp32 level, as single ops: fp32 tensors (NHWC; the thin output conv writes NCHW),
        // filter packed by dmme_pack_weight(DMME_F16R32): [Cout][taps][Cin / 32][hi 32 | lo 32] halves.  No fall-back to other kernels.
        a.mix = a.out_nchw ? 3 : a.taps == 1 ? 4 : 1;
        a.f16 = 1;
        if (a.mix == 4) {
            DMME_REQUIRE(conv1x1_pipe_supported(DMME_F16, a), DMME_ERR_UNSUPPORTED, "conv2d(fp16r32): the split-pass 1x1 kernel does not take this shape");
            return launch_conv1x1_pipe(DMME_F16, a, (hipStream_t)stream);
        }
        if (a.mix == 3) {
            DMME_REQUIRE(conv_out_thin_supported(DMME_F16, a), DMME_ERR_UNSUPPORTED, "conv2d(fp16r32): the thin output conv does not take this shape");
            return launch_conv_out_thin(a, (hipStream_t)stream);
        }
        DMME_REQUIRE(conv_pipe_supported(DMME_F16, a), DMME_ERR_UNSUPPORTED, "conv2d(fp16r32): the split-pass 3x3 kernel does not take this shape");
        return launch_conv_pipe(DMME_F16, a, (hipStream_t)stream);
    }
    const int dt = a.x3 ? DMME_F32 : d->dtype;
    if (d->force_generic == 2 && conv_mfma_supported(dt, a)) return launch_conv_mfma(dt, a, (hipStream_t)stream);
    if (!d->force_generic && conv_out_thin_supported(dt, a)) return launch_conv_out_thin(a, (hipStream_t)stream);
    if (!d->force_generic && conv1x1_pipe_supported(dt, a)) return launch_conv1x1_pipe(dt, a, (hipStream_t)stream);
    if (!d->force_generic && conv_pipe_supported(dt, a)) return launch_conv_pipe(dt, a, (hipStream_t)stream);
    if (!d->force_generic && conv_mfma_supported(dt, a)) return launch_conv_mfma(dt, a, (hipStream_t)stream);
    return launch_conv_generic(dt, a, (hipStream_t)stream);
}

DMME_API int dmme_conv2d_res(const dmme_conv_desc* d, const void* src1, const void* src2, const void* weight, const float* bias, const float* scale,
                             const float* shift, const float* dmask, const void* r_src1, const void* r_src2, int r_C1, int r_C2, const void* r_weight,
                             const float* r_bias, void* dst, void* stream) {
    DMME_REQUIRE(d && src1 && weight && bias && dst && r_src1 && r_weight && r_bias, DMME_ERR_INVALID, "conv2d_res: null argument");
    DMME_REQUIRE(d->dtype == DMME_BF16 || d->dtype == DMME_F16, DMME_ERR_UNSUPPORTED, "conv2d_res: 16-bit tensors only (dtype %d)", d->dtype);
    ConvArgs a{};
    a.src1 = src1; a.src2 = src2; a.w = weight; a.bias = bias; a.scale = scale; a.shift = shift; a.dmask = dmask; a.dst = dst;
    a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = src2 ? d->C2 : 0;
    a.up = d->upsample; a.stride = d->stride; a.taps = d->taps;
    a.Hout = a.Hin; a.Wout = a.Win; a.Cout = d->Cout;
    a.pro_silu = d->pro_silu; a.out_silu = d->out_silu;
    a.f16 = d->dtype == DMME_F16;
    a.stamps = g_stamps;
    a.r_src1 = r_src1; a.r_src2 = r_C2 > 0 ? r_src2 : nullptr; a.r_C1 = r_C1; a.r_C2 = r_C2 > 0 ? r_C2 : 0; a.r_w = r_weight; a.r_bias = r_bias;
    DMME_REQUIRE(a.taps == 9 && a.stride == 1 && !a.up && conv_pipe_rseg_supported(d->dtype, a), DMME_ERR_UNSUPPORTED,
                 "conv2d_res: the wave-specialised 3x3 kernel does not take this shape with a residual segment (%d+%d -> %d channels, %d+%d raw, %dx%dx%d)", a.C1, a.C2,
                 a.Cout, a.r_C1, a.r_C2, a.N, a.Hin, a.Win);
    return launch_conv_pipe(d->dtype, a, (hipStream_t)stream);
}

DMME_API int dmme_groupnorm_scale_shift(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2, int groups,
                               const float* gamma, const float* beta, float eps, float* scale, float* shift,
                               float* partial_scratch, int force_generic, void* stream) {
    DMME_REQUIRE(src1 && gamma && beta && scale && shift, DMME_ERR_INVALID, "groupnorm: null argument");
    if (!src2) C2 = 0;
    DMME_REQUIRE(groups > 0 && (C1 + C2) % groups == 0, DMME_ERR_INVALID, "groupnorm: %d channels not divisible by %d groups",
                 C1 + C2, groups);
    if (!force_generic && partial_scratch && gn_fast_supported(dtype, N, HW, C1, C2, groups))
        return launch_gn_fast(dtype, src1, src2, N, HW, C1, C2, groups, gamma, beta, eps, scale, shift, nullptr,
                              partial_scratch, (hipStream_t)stream);
    return launch_gn_generic(dtype, src1, src2, N, HW, C1, C2, groups, gamma, beta, eps, scale, shift, nullptr,
                             (hipStream_t)stream);
}

DMME_API int dmme_attention(int dtype, const void* qkv, int N, int S, int C, void* out, int force_generic, void* stream) {
    DMME_REQUIRE(qkv && out && N > 0 && S > 0 && C > 0, DMME_ERR_INVALID, "attention: bad argument");
    if (dtype == DMME_BF16X3) {  // fp32 buffers; the three-pass MFMA kernel where it applies
        if (!force_generic && attn_x3_supported(N, S, C, 1)) return launch_attn_x3(qkv, N, S, C, 1, out, (hipStream_t)stream);
        dtype = DMME_F32;
    }
    if (!force_generic && attn_mfma_supported(dtype, N, S, C)) return launch_attn_mfma(dtype, qkv, N, S, C, out, nullptr, (hipStream_t)stream);
    return launch_attn_generic(dtype, qkv, N, S, C, out, (hipStream_t)stream);
}

DMME_API int dmme_attention_proj(int dtype, const void* qkv, int N, int S, int C, const void* w, const float* bias, const void* res, void* dst, void* ctx,
                                 float* gn_part, int gn_cg, void* stream) {
    DMME_REQUIRE(qkv && w && bias && res && dst && N > 0 && S > 0 && C > 0, DMME_ERR_INVALID, "attention_proj: bad argument");
    return launch_attn_proj(dtype, qkv, N, S, C, ctx, nullptr, w, bias, res, dst, gn_part, S / 32, gn_cg, (hipStream_t)stream, g_stamps);
}

DMME_API int dmme_attention_heads(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, int force_generic, void* stream) {
    DMME_REQUIRE(qkv && out && N > 0 && S > 0 && C > 0 && heads > 0 && C % heads == 0, DMME_ERR_INVALID, "attention_heads: bad argument");
    if (dtype == DMME_BF16X3) {
        if (!force_generic && attn_x3_supported(N, S, C, heads)) return launch_attn_x3(qkv, N, S, C, heads, out, (hipStream_t)stream);
        dtype = DMME_F32;
    }
    if (!force_generic && attn_heads_mfma_supported(dtype, N, S, C, heads))
        return launch_attn_heads_mfma(dtype, qkv, N, S, C, heads, out, nullptr, (hipStream_t)stream);
    return launch_attn_heads(dtype, qkv, N, S, C, heads, out, (hipStream_t)stream);
}

DMME_API int dmme_nchw_to_nhwc(int dtype, const float* src, int N, int C, int HW, void* dst, void* stream) {
    DMME_REQUIRE(src && dst, DMME_ERR_INVALID, "nchw_to_nhwc: null argument");
    return launch_nchw_to_nhwc(dtype, src, N, C, HW, dst, (hipStream_t)stream);
}
DMME_API int dmme_nhwc_to_nchw(int dtype, const void* src, int N, int C, int HW, float* dst, void* stream) {
    DMME_REQUIRE(src && dst, DMME_ERR_INVALID, "nhwc_to_nchw: null argument");
    return launch_nhwc_to_nchw(dtype, src, N, C, HW, dst, (hipStream_t)stream);
}
DMME_API int dmme_pack_weight(int dtype, const float* src, int Cout, int Cin, int taps, void* dst, void* stream) {
    DMME_REQUIRE(src && dst, DMME_ERR_INVALID, "pack_weight: null argument");
    return launch_pack_weight(dtype, src, Cout, Cin, taps, dst, (hipStream_t)stream);
}

DMME_API int dmme_event_create(void** ev) {
    DMME_REQUIRE(ev, DMME_ERR_INVALID, "event_create: null");
    hipEvent_t e;
    DMME_CHECK_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return DMME_OK;
}
DMME_API int dmme_event_record(void* ev, void* stream) {
    DMME_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return DMME_OK;
}
DMME_API int dmme_event_elapsed_ms(void* start, void* stop, float* ms) {
    DMME_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    DMME_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return DMME_OK;
}
DMME_API int dmme_event_destroy(void* ev) {
    DMME_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return DMME_OK;
}

}  // extern "C"
