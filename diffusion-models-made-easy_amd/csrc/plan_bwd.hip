// Backward side of the plan: the grouped weight-gradient tables, the reverse walk over the op list (data gradients through the
// forward kernels, GroupNorm / attention / time-MLP backward), gradient buckets for the overlapped data-parallel exchange.
#include "plan.h"

using namespace dmme;

namespace dmme {

// Grouped weight gradients: every 3x3 stride-1 conv the all-taps MFMA kernel supports is taken out of the per-layer
// sequence; its (cout tile, cin tile) pairs are cut into jobs of at most `q` consecutive 64-pixel tiles, longest first.
void build_wgrad_group(dmme_plan* P, dmme_plan::WgGroup& G, int gi, int op_lo, int op_hi) {
    const int taps = gi == 1 ? 1 : 9, stride = gi == 2 ? 2 : 1;
    G.taps = taps;
    G.stride = stride;
    if (stride == 2 && debug_route("no_wg_s2")) return;
    if (!is16(P->dtype) || getenv("DMME_NO_WGRAD_GROUP")) return;
    // (the stride-2 table is three small layers: shorter jobs, or 80 workgroups would carry it)
    const int q = stride == 2 ? 16 : 64;
    struct Grp { int layer, n_co, n_ci, tile0, ntiles; };
    std::vector<Grp> groups;
    for (int oi = (int)P->ops.size() - 1; oi >= 0; --oi) {
        Op& o = P->ops[oi];
        if (o.kind != OP_CONV || o.src1 < 0 || o.dst < 0 || o.taps != taps || o.stride != stride || oi < op_lo || oi >= op_hi) continue;
        ConvArgs a{};
        fill_conv(P, o, nullptr, nullptr, nullptr, nullptr, nullptr, 1, a);
        WgLayer L{};
        int CO = 0, CI = 0;
        if (!wgrad_mfma_supported(P->dtype, a) || !wgrad_group_layer(P->dtype, a, L, &CO, &CI)) continue;
        L.src1_off = P->tensors[o.src1].off;
        L.src2_off = o.src2 >= 0 ? P->tensors[o.src2].off : -1;
        L.scale_off = o.gn >= 0 ? P->ops[o.gn].gn_scale : -1;
        L.shift_off = o.gn >= 0 ? P->ops[o.gn].gn_shift : -1;
        L.dmask_off = o.dmask_off;
        // The weight gradient's second operand is the conv's ACTIVATED input.  Recomputing GroupNorm + SiLU + dropout per MFMA operand
        // made the grouped kernel VALU-issue bound; instead it reads the activated tensor: the forward's own (small maps, use_act), or
        // one the GroupNorm backward of this conv writes on its way (it holds x, scale, shift and the mask anyway: one more store).
        L.act_off = -1;
        L.act_bws = 0;
        if (o.gn >= 0 && !debug_route("no_wg_act")) {
            const Op& gop = P->ops[o.gn];
            if (o.use_act && gop.gn_act >= 0) {
                L.act_off = gop.gn_act;
            } else if (gop.gn_src1 == o.src1 && gop.gn_src2 == o.src2 && gn_bwd_fast_supported(P->dtype, a.Hin * a.Win, a.C1, a.C2)) {
                if (o.wg_act < 0) {
                    o.wg_act = align_up(P->bws_bytes, 256);
                    P->bws_bytes = o.wg_act + (int64_t)P->B * a.Hin * a.Win * (a.C1 + a.C2) * (int64_t)dtype_size(P->dtype);
                }
                L.act_off = o.wg_act;
                L.act_bws = 1;
            }
        }
        L.dy_off = P->gt_off[o.dst];
        L.dw_off = P->params[o.w].wp_off;
        o.wg_layer = (int)G.layers.size();
        G.layers.push_back(L);
        const int n_co = (L.Cout + CO - 1) / CO, n_ci = (L.C1 + L.C2) / CI;
        const int ns = (L.g.tiles_m + q - 1) / q;
        for (int sp = 0; sp < ns; ++sp) {
            Grp gr{};
            gr.layer = o.wg_layer;
            gr.n_co = n_co;
            gr.n_ci = n_ci;
            gr.tile0 = (int)((int64_t)L.g.tiles_m * sp / ns);
            gr.ntiles = (int)((int64_t)L.g.tiles_m * (sp + 1) / ns) - gr.tile0;
            if (gr.ntiles > 0) groups.push_back(gr);
        }
    }
    G.dma = !G.layers.empty() && !debug_route("no_wg_dma");
    for (const WgLayer& L : G.layers)
        if (!(L.act_off >= 0 || ((L.C2 == 0 || taps == 1) && L.scale_off < 0 && L.dmask_off < 0 && !L.pro_silu)) || L.Cout % (taps == 9 ? 64 : 128) ||
            (taps == 1 && (L.C1 + L.C2) % 128))
            G.dma = 0;
    if (stride == 2) {
        if (!G.dma) {  // no register-staged fallback for stride 2: those layers stay on the per-layer kernel
            for (Op& o : P->ops)
                if (o.kind == OP_CONV && o.taps == 9 && o.stride == 2) o.wg_layer = -1;
            G.layers.clear();
            return;
        }
        G.dma = 2;
    }
    // All (cout tile, cin tile) jobs of one pixel range read the same dY and activation tiles: they go to ONE XCD
    // (consecutive positions of its round-robin slice of the grid, blockIdx % 8), so the re-reads hit that XCD's L2
    // instead of HBM.  Groups are placed longest first on the least-loaded XCD; short slices are padded with empty jobs.
    std::stable_sort(groups.begin(), groups.end(), [](const Grp& x, const Grp& y) { return x.ntiles > y.ntiles; });
    const int NX = 8;
    std::vector<WgJob> lists[NX];
    int64_t load[NX] = {0};
    for (const Grp& gr : groups) {
        int best = 0;
        for (int x = 1; x < NX; ++x)
            if (load[x] < load[best]) best = x;
        for (int cot = 0; cot < gr.n_co; ++cot)
            for (int cit = 0; cit < gr.n_ci; ++cit) {
                WgJob j{};
                j.layer = gr.layer;
                j.cot = cot;
                j.cit = cit;
                j.tile0 = gr.tile0;
                j.ntiles = gr.ntiles;
                lists[best].push_back(j);
            }
        load[best] += (int64_t)gr.ntiles * gr.n_co * gr.n_ci;
    }
    size_t longest = 0;
    for (int x = 0; x < NX; ++x) longest = std::max(longest, lists[x].size());
    for (size_t pos = 0; pos < longest; ++pos)
        for (int x = 0; x < NX; ++x) {
            WgJob j{};
            if (pos < lists[x].size()) j = lists[x][pos];
            G.jobs.push_back(j);
        }
}

}  // namespace dmme

extern "C" {

DMME_API int64_t dmme_unet_plan_packed_bwd_bytes(const dmme_plan* plan) { return plan ? plan->packed_bwd_bytes : 0; }
DMME_API int64_t dmme_unet_plan_bwd_workspace_bytes(const dmme_plan* plan) { return plan ? plan->bws_bytes : 0; }

DMME_API int dmme_unet_pack_params_bwd(const dmme_plan* plan, const float* ref_flat, void* packed_bwd, void* stream) {
    DMME_REQUIRE(plan && ref_flat && packed_bwd, DMME_ERR_INVALID, "pack_params_bwd: null argument");
    DMME_REQUIRE(plan->items_bwd_dev, DMME_ERR_INVALID, "pack_params_bwd: plan was created without a device");
    return launch_pack_table(plan->dtype, plan->items_bwd_dev, plan->n_items_bwd, ref_flat, packed_bwd, (hipStream_t)stream);
}

static int backward_impl(const dmme_plan* plan, const void* packed, const void* packed_bwd, const float* x, const int64_t* t, int t_len,
                         const float* d_y, void* workspace, void* bwd_workspace, const float* drop_masks, float* grad_flat, float* d_x,
                         void* stream, dmme_bucket_fn ready, void* user) {
    DMME_REQUIRE(plan && packed && packed_bwd && x && t && d_y && workspace && bwd_workspace && grad_flat, DMME_ERR_INVALID,
                 "unet_backward: null argument");
    DMME_REQUIRE(t_len == 1 || t_len == plan->B, DMME_ERR_INVALID, "unet_backward: bad t_len %d", t_len);
    if (int rc0 = lvl_check(plan, "unet_backward", (hipStream_t)stream, true)) return rc0;  // (the forward this backward differentiates ran through the engine)
    DMME_REQUIRE(!plan->mix, DMME_ERR_UNSUPPORTED, "unet_backward: precision fp16r32 is an inference mode");
    DMME_REQUIRE(plan->nograd_ws != workspace, DMME_ERR_INVALID,
                 "unet_backward: the last forward into this workspace was dmme_unet_forward_nograd / dmme_chain_step, which leave out the tensors only a "
                 "backward pass reads; run dmme_unet_forward first");
    const dmme_plan* P = plan;
    hipStream_t s = (hipStream_t)stream;
    const char* pk = (const char*)packed;
    const char* pkb = (const char*)packed_bwd;
    char* ws = (char*)workspace;
    char* bws = (char*)bwd_workspace;
    const int B = P->B, dt = P->dtype, nt = t_len, G = P->cfg.num_groups;
    std::vector<char> written(P->tensors.size(), 0);
    auto gptr = [&](int id) -> char* { return bws + P->gt_off[id]; };
    auto claim = [&](int id) -> int {  // 0: first contribution (write), 1: accumulate
        const int acc = written[id];
        written[id] = 1;
        return acc;
    };
    // identity-residual branches (d x += d out of a ResBlock / attention block) are not launched on their own: the pointer waits here
    // until the GroupNorm backward that writes x's gradient anyway (norm1 / the attention norm of the same block) takes it as one more
    // addend; anything else that needs x's gradient first gets it through flush_pending
    const bool res_extra_off = (debug_route("no_res_extra") != 0);
    std::vector<const char*> pending(P->tensors.size(), nullptr);
    auto flush_pending = [&](int id) -> int {
        if (id < 0 || !pending[id]) return DMME_OK;
        const Tensor& t = P->tensors[id];
        const char* src = pending[id];
        pending[id] = nullptr;
        const int acc = claim(id);
        return launch_grad_acc(dt, src, gptr(id), nullptr, t.C, 0, acc, 0, 0, B, t.H, t.W, s);
    };

    DMME_CHECK_HIP(hipMemsetAsync(bws + P->bws_zero, 0, (size_t)P->bws_zero_bytes, s));
    float* wimage = (float*)(bws + P->bws_wimage);
    float* dtproj = (float*)(bws + P->bws_dtproj);
    char* tmp = bws + P->bws_tmp;
    int rc = launch_nchw_to_nhwc(dt, d_y, B, P->out_channels, P->H * P->W, bws + P->bws_dy, s);
    if (rc != DMME_OK) return rc;
    const bool buckets = ready != nullptr && !P->gb.empty();  // bucketed mode: deferred work flushed per gradient bucket
    const int emb = P->cfg.emb_dim, pos = P->cfg.pos_dim, tc = P->tproj_cols;
    const float* temb = (const float*)(ws + P->ws_temb);
    // deferred launches of one gradient bucket (b >= 0) or of everything (b = -1): bias + time rows, grouped weight gradients, unpack,
    // the per-block time-projection weight gradients
    auto flush = [&](int b) -> int {
        int r = DMME_OK;
        const dmme_plan::GradBucket* GBk = b >= 0 ? &P->gb[b] : nullptr;
        if (P->bias_jobs_dev && P->col_jobs_dev) {
            const int j0 = GBk ? GBk->col0 : 0, j1 = GBk ? GBk->col1 : (int)P->col_jobs.size();
            if (j1 > j0) r = launch_colsum_group(dt, P->col_jobs_dev + j0, j1 - j0, bws, B, s);
            if (r != DMME_OK) return r;
        }
        if (P->bias_jobs_dev) {
            const int j0 = GBk ? GBk->bias0 : 0, j1 = GBk ? GBk->bias1 : (int)P->bias_jobs.size();
            if (j1 > j0) r = launch_bias_tproj_group(P->bias_jobs_dev + j0, j1 - j0, bws, grad_flat, dtproj, B, tc, nt, s);
            if (r != DMME_OK) return r;
        }
        for (int k = 0; k < 3; ++k) {
            const dmme_plan::WgGroup& G = GBk ? GBk->wg[k] : P->wg[k];
            if (!G.jobs_dev) continue;
            r = launch_wgrad_group(dt, G.taps, G.layers_dev, G.jobs_dev, (int)G.jobs.size(), ws, bws, drop_masks, wimage, s, G.dma, bws + P->bws_zpage);
            if (r != DMME_OK) return r;
        }
        {
            std::vector<std::pair<int, int>> all_items{{0, P->n_items_unpack}};
            for (const auto& ir : (GBk ? GBk->unpack : all_items)) {
                if (ir.second > ir.first) r = launch_wgrad_unpack(P->items_unpack_dev + ir.first, ir.second - ir.first, wimage, grad_flat, s);
                if (r != DMME_OK) return r;
            }
        }
        std::vector<std::pair<int, int>> all_cols{{0, tc}};
        for (const auto& cr : (GBk ? GBk->tcols : all_cols)) {
            const int c0 = cr.first, c1 = cr.second;
            if (c1 <= c0) continue;
            if (P->tp_tiles_dev) {  // every block's dW / db in one launch each
                r = launch_small_gemm_tn_tiled(dtproj + c0, tc, temb, emb, c1 - c0, emb, nt, grad_flat, emb, P->tp_tiles_dev + c0 / 64, s);
                if (r == DMME_OK) r = launch_nsum_tiled(dtproj + c0, nt, c1 - c0, tc, 1, grad_flat, P->tp_tiles_dev + P->tp_n64 + c0 / 32, s);
                if (r != DMME_OK) return r;
            } else {
                for (const auto& tb : P->tblocks) {
                    if (tb.col < c0 || tb.col >= c1) continue;
                    // dW_block[o][k] += sum_r dtproj[r][col+o] temb[r][k];  db_block[o] += sum_r dtproj[r][col+o]
                    r = launch_small_gemm(dt, 2, dtproj + tb.col, tc, temb, emb, tb.cout, emb, nt, nullptr, 0, grad_flat + P->params[tb.tw].ref_off, emb, s);
                    if (r == DMME_OK) r = launch_nsum(dtproj + tb.col, nt, tb.cout, tc, 1, grad_flat + P->params[tb.tb].ref_off, s);
                    if (r != DMME_OK) return r;
                }
            }
        }
        return r;
    };
    auto hand_over = [&](int b) {
        for (const auto& r : P->gb[b].ranges) ready(user, b, r.first, r.second);
    };
    int next_bucket = 0;  // the bucket whose stretch the reverse walk is in
    if (P->cfg.arch == DMME_ARCH_IDDPM && nt == 1)  // shared timestep row: the GroupNorm backward accumulates into it atomically
        DMME_CHECK_HIP(hipMemsetAsync(dtproj, 0, (size_t)P->tproj_cols * 4, s));

    for (int oi = (int)P->ops.size() - 1; oi >= 0 && rc == DMME_OK; --oi) {
        if (buckets && next_bucket + 1 < (int)P->gb.size() && oi == P->gb[next_bucket].op_lo - 1) {
            // every op of this bucket has run (a pending identity-residual gradient that belongs to a tensor of the NEXT stretch stays
            // pending: it carries no parameter gradient): finish the bucket's parameter gradients and hand it to the exchange
            rc = flush(next_bucket);
            if (rc != DMME_OK) break;
            hand_over(next_bucket);
            ++next_bucket;
        }
        const Op& o = P->ops[oi];
        if (o.kind == OP_ATTN) {
            rc = flush_pending(o.at_out);
            if (rc != DMME_OK) break;
            const Tensor& q = P->tensors[o.at_qkv];
            const int S = q.H * q.W, C = q.C / 3;
            DMME_REQUIRE(written[o.at_out], DMME_ERR_INVALID, "backward: attention output has no gradient");
            if (o.at_heads > 1 && attn_heads_mfma_supported(dt, B, S, C, o.at_heads))
                rc = launch_attn_heads_bwd_mfma(dt, ws + q.off, ws + P->tensors[o.at_out].off, gptr(o.at_out), (const float*)(ws + o.at_lse), B, S, C,
                                                o.at_heads, bws + P->bws_attP, bws + P->bws_attdS, gptr(o.at_qkv), s);
            else if (o.at_heads > 1)
                rc = launch_attn_heads_bwd(dt, ws + q.off, gptr(o.at_out), B, S, C, o.at_heads, (float*)(bws + P->bws_attP),
                                           (float*)(bws + P->bws_attdS), gptr(o.at_qkv), s);
            else if (attn_bwd_mfma_supported(dt, B, S, C))
                rc = launch_attn_bwd_mfma(dt, ws + q.off, ws + P->tensors[o.at_out].off, gptr(o.at_out), (const float*)(ws + o.at_lse), B, S, C,
                                          bws + P->bws_attP, bws + P->bws_attdS, gptr(o.at_qkv), s);
            else
                rc = launch_attn_bwd_generic(dt, ws + q.off, gptr(o.at_out), B, S, C, (float*)(bws + P->bws_attP),
                                             (float*)(bws + P->bws_attdS), gptr(o.at_qkv), s);
            written[o.at_qkv] = 1;
            continue;
        }
        if (o.kind != OP_CONV) continue;
        ConvArgs a{};
        fill_conv(P, o, pk, x, nullptr, ws, drop_masks, nt, a);
        if (o.dst >= 0) {
            rc = flush_pending(o.dst);
            if (rc != DMME_OK) break;
        }
        const char* dy = o.dst == -2 ? bws + P->bws_dy : gptr(o.dst);
        if (o.dst != -2) DMME_REQUIRE(written[o.dst], DMME_ERR_INVALID, "backward: tensor %d has no gradient", o.dst);
        const int Cin = a.C1 + a.C2;
        float* rowsum = (float*)(bws + o.b_rowsum);
        // 1. bias and time-embedding-row gradients (column sums of dY)
        if (o.bias_deferred && P->bias_jobs_dev && P->col_jobs_dev)
            rc = DMME_OK;  // its column sums come from the grouped launch of the flush
        else if (o.bias_deferred && P->bias_jobs_dev)
            rc = launch_colsum_fast(dt, dy, B, a.Hout * a.Wout, a.Cout, rowsum, nullptr, nullptr, P->tproj_cols, nt, s);
        else if (colsum_fast_supported(dt, a.Hout * a.Wout, a.Cout))
            rc = launch_colsum_fast(dt, dy, B, a.Hout * a.Wout, a.Cout, rowsum, grad_flat + P->params[o.b].ref_off,
                                    o.tproj_col >= 0 ? dtproj + o.tproj_col : nullptr, P->tproj_cols, nt, s);
        else
            rc = launch_colsum(dt, dy, B, a.Hout * a.Wout, a.Cout, rowsum, grad_flat + P->params[o.b].ref_off,
                               o.tproj_col >= 0 ? dtproj + o.tproj_col : nullptr, P->tproj_cols, nt, s);
        if (rc != DMME_OK) break;
        // 2. weight gradient: deferred to the grouped launch below, or per layer (packed image / reference layout)
        if (o.wg_layer >= 0 && (buckets ? P->gb[next_bucket].wg[wg_index(o)].jobs_dev : P->wg[wg_index(o)].jobs_dev))
            rc = DMME_OK;
        else if (wgrad_mfma_supported(dt, a))
            rc = launch_wgrad_mfma(dt, a, dy, wimage + P->params[o.w].wp_off, s);
        else if (wgrad_small_supported(dt, a))
            rc = launch_wgrad_small(dt, a, dy, grad_flat + P->params[o.w].ref_off, s);
        else
            rc = launch_wgrad_generic(dt, a, dy, grad_flat + P->params[o.w].ref_off, s);
        if (rc != DMME_OK) break;
        // 3. data gradient: the forward kernel on dY with transposed, tap-flipped weights
        if (o.src1 >= 0) {
            ConvArgs d{};
            d.src1 = dy;
            d.C1 = a.Cout;
            d.N = B;
            d.Hin = a.Hout;
            d.Win = a.Wout;
            d.up = o.stride == 2 ? 2 : 0;
            d.stride = 1;
            d.taps = o.taps;
            d.Hout = d.up ? 2 * d.Hin : d.Hin;
            d.Wout = d.up ? 2 * d.Win : d.Win;
            d.Cout = Cin;
            d.w = pkb + P->params[o.w].packed_bwd_off;
            d.dst = tmp;
            d.x3 = P->x3;
        d.f16 = P->dtype == DMME_F16;  // (launchers without a dtype argument: conv1x1_as, the thin output conv)
            d.f16 = P->dtype == DMME_F16;  // (launchers without a dtype argument: conv1x1_as, the thin output conv)
            if (P->splitk_floats > 0) {
                d.splitk = (float*)(ws + P->ws_splitk);
                d.splitk_cap = P->splitk_floats;
            }
            const Tensor& t1 = P->tensors[o.src1];
            char* g1 = gptr(o.src1);
            char* g2 = o.src2 >= 0 ? gptr(o.src2) : nullptr;
            const char* extra = nullptr;  // a waiting residual branch of the source: taken along by the GroupNorm backward below
            if (pending[o.src1]) {
                if (o.gn >= 0 && o.src2 < 0 && gn_bwd_fast_supported(dt, t1.H * t1.W, a.C1, a.C2)) {
                    extra = pending[o.src1];
                    pending[o.src1] = nullptr;
                } else {
                    rc = flush_pending(o.src1);
                    if (rc != DMME_OK) break;
                }
            }
            if (o.src2 >= 0) {
                rc = flush_pending(o.src2);
                if (rc != DMME_OK) break;
            }
            const int acc1 = claim(o.src1), acc2 = o.src2 >= 0 ? claim(o.src2) : 0;
            // a conv with no norm in front of it, one source and no fused upsample: its data gradient IS the source's gradient -
            // written (or, through the epilogue's residual input, accumulated in place: each vector is read and written by one thread)
            // straight into that buffer instead of a scratch tensor plus an accumulation launch
            const bool direct_off = (debug_route("no_dgrad_direct") != 0);
            const bool dgrad_direct = !direct_off && o.gn < 0 && o.src2 < 0 && o.up != 1;
            if (dgrad_direct) {
                d.dst = g1;
                if (acc1) {
                    d.res1 = g1;
                    d.R1 = Cin;
                }
            }
            rc = run_any_conv(dt, d, s);
            if (rc != DMME_OK) break;

            if (o.gn >= 0) {
                const Op& gop = P->ops[o.gn];
                GnMod mod{};
                if (gop.gn_mod_col >= 0) {  // scale-shift conditioning: effective gamma + gradients of the (shift | scale) projection rows
                    mod.t_scale = (const float*)(ws + P->ws_tproj) + gop.gn_mod_col + gop.gn_mod_C;
                    mod.beta = (const float*)(pk + P->params[gop.gn_beta].packed_off);
                    mod.d_shift = dtproj + gop.gn_mod_col;
                    mod.d_scale = dtproj + gop.gn_mod_col + gop.gn_mod_C;
                    mod.ld = P->tproj_cols;
                    mod.nt = nt;
                }
                if (gn_bwd_fast_supported(dt, t1.H * t1.W, a.C1, a.C2))
                    rc = launch_gn_bwd_fast(dt, tmp, a.src1, a.src2, B, t1.H * t1.W, a.C1, a.C2, G,
                                            (const float*)(pk + P->params[gop.gn_gamma].packed_off), (const float*)(ws + gop.gn_mr), a.scale,
                                            a.shift, a.dmask, a.pro_silu, g1, g2, acc1, acc2, grad_flat + P->params[gop.gn_gamma].ref_off,
                                            grad_flat + P->params[gop.gn_beta].ref_off, (float*)(bws + o.b_ab), (float*)(bws + P->bws_gnS), mod, s,
                                            o.wg_act >= 0 ? bws + o.wg_act : nullptr,
                                            o.gn_rows_deferred && P->bias_jobs_dev ? (float*)(bws + o.b_gnrows) : nullptr, extra);
                else
                rc = launch_gn_bwd_generic(dt, tmp, a.src1, a.src2, B, t1.H * t1.W, a.C1, a.C2, G,
                                           (const float*)(pk + P->params[gop.gn_gamma].packed_off), (const float*)(ws + gop.gn_mr),
                                           a.scale, a.shift, a.dmask, a.pro_silu, g1, g2, acc1, acc2,
                                           grad_flat + P->params[gop.gn_gamma].ref_off, grad_flat + P->params[gop.gn_beta].ref_off, mod, s);
            } else if (!dgrad_direct) {
                rc = launch_grad_acc(dt, tmp, g1, g2, a.C1, a.C2, acc1, acc2, o.up == 1 ? 1 : 0, B, t1.H, t1.W, s);
            }
            if (rc != DMME_OK) break;
        }
        if (o.src1 == -2 && d_x) {  // gradient with respect to the network input (NCHW fp32), only on request
            ConvArgs d{};
            d.src1 = dy;
            d.C1 = a.Cout;
            d.N = B;
            d.Hin = d.Hout = a.Hout;
            d.Win = d.Wout = a.Wout;
            d.stride = 1;
            d.taps = o.taps;
            d.Cout = Cin;
            d.w = pkb + P->params[o.w].packed_bwd_off;
            d.dst = d_x;
            d.out_nchw = 1;
            d.x3 = P->x3;
        d.f16 = P->dtype == DMME_F16;  // (launchers without a dtype argument: conv1x1_as, the thin output conv)
            d.f16 = P->dtype == DMME_F16;  // (launchers without a dtype argument: conv1x1_as, the thin output conv)
            rc = conv_mfma_supported(dt, d) ? launch_conv_mfma(dt, d, s) : launch_conv_generic(dt, d, s);
            if (rc != DMME_OK) break;
        }
        // 4. residual branch: d(res) += dY
        if (o.res1 >= 0 && o.res_alias) {
            written[o.res1] = 1;  // (its gradient buffer is dY itself)
        } else if (o.res1 >= 0) {
            const int R1 = P->tensors[o.res1].C;
            if (!res_extra_off && o.res2 < 0 && R1 == a.Cout && o.dst >= 0) {
                rc = flush_pending(o.res1);  // (one waiting branch per tensor)
                pending[o.res1] = dy;
            } else {
                const int acc1 = claim(o.res1), acc2 = o.res2 >= 0 ? claim(o.res2) : 0;
                rc = launch_grad_acc(dt, dy, gptr(o.res1), o.res2 >= 0 ? gptr(o.res2) : nullptr, R1, a.Cout - R1, acc1, acc2, 0, B,
                                     a.Hout, a.Wout, s);
            }
        }
    }
    for (int id = 0; id < (int)pending.size() && rc == DMME_OK; ++id) rc = flush_pending(id);
    if (rc != DMME_OK) return rc;
    rc = flush(buckets ? (int)P->gb.size() - 1 : -1);
    if (rc != DMME_OK) return rc;

    // ---- time MLP backward (models/ddpm.py:211-217 and the per-block Linear at :101-104) ----
    const float* h1 = (const float*)(ws + P->ws_th1);
    const float* esin = (const float*)(ws + P->ws_tsin);
    float* dtemb = (float*)(bws + P->bws_dtemb);
    float* dh1 = (float*)(bws + P->bws_dh1);
    float* z = (float*)(bws + P->bws_z);
    // input gradients of the Linears as NT GEMMs against a transposed copy of the weights (K contiguous in both operands)
    char* wT = bws + P->bws_wT;
    rc = launch_transpose(dt, pk + P->tproj_w_off, tc, emb, wT, s);
    if (rc == DMME_OK) rc = launch_small_gemm(dt, 0, dtproj, tc, wT, tc, nt, emb, tc, nullptr, 0, dtemb, emb, s);
    if (rc != DMME_OK) return rc;
    // temb = silu(z2), z2 = h1 W2^T + b2
    // (the forward kept both pre-activations when it ran at a training batch: no recompute GEMMs here)
    const bool saved_pre = nt > 4 && P->ws_tz1 >= 0 && P->ws_tz2 >= 0 && !debug_route("no_time_pre");
    if (saved_pre)
        z = (float*)(ws + P->ws_tz2);
    else
        rc = launch_small_gemm(dt, 0, h1, emb, pk + P->params[P->p_l2w].packed_off, emb, nt, emb, emb, (const float*)(pk + P->params[P->p_l2b].packed_off), 0, z, emb, s);
    if (rc == DMME_OK) rc = launch_silu_bwd(dtemb, z, nt * emb, s);
    if (rc == DMME_OK) rc = launch_small_gemm(dt, 2, dtemb, emb, h1, emb, emb, emb, nt, nullptr, 0, grad_flat + P->params[P->p_l2w].ref_off, emb, s);
    if (rc == DMME_OK) rc = launch_nsum(dtemb, nt, emb, emb, 1, grad_flat + P->params[P->p_l2b].ref_off, s);
    if (rc == DMME_OK) rc = launch_transpose(dt, pk + P->params[P->p_l2w].packed_off, emb, emb, wT, s);
    if (rc == DMME_OK) rc = launch_small_gemm(dt, 0, dtemb, emb, wT, emb, nt, emb, emb, nullptr, 0, dh1, emb, s);
    // h1 = silu(z1), z1 = e W1^T + b1
    if (saved_pre)
        z = (float*)(ws + P->ws_tz1);
    else if (rc == DMME_OK)
        rc = launch_small_gemm(dt, 0, esin, pos, pk + P->params[P->p_l1w].packed_off, pos, nt, emb, pos, (const float*)(pk + P->params[P->p_l1b].packed_off), 0, z, emb, s);
    if (rc == DMME_OK) rc = launch_silu_bwd(dh1, z, nt * emb, s);
    if (rc == DMME_OK) rc = launch_small_gemm(dt, 2, dh1, emb, esin, pos, emb, pos, nt, nullptr, 0, grad_flat + P->params[P->p_l1w].ref_off, pos, s);
    if (rc == DMME_OK) rc = launch_nsum(dh1, nt, emb, emb, 1, grad_flat + P->params[P->p_l1b].ref_off, s);
    if (rc == DMME_OK && buckets) hand_over((int)P->gb.size() - 1);
    return rc;
}

DMME_API int dmme_unet_backward(const dmme_plan* plan, const void* packed, const void* packed_bwd, const float* x,
                                const int64_t* t, int t_len, const float* d_y, void* workspace, void* bwd_workspace,
                                const float* drop_masks, float* grad_flat, float* d_x, void* stream) {
    return backward_impl(plan, packed, packed_bwd, x, t, t_len, d_y, workspace, bwd_workspace, drop_masks, grad_flat, d_x, stream, nullptr, nullptr);
}

DMME_API int dmme_unet_backward_buckets(const dmme_plan* plan, const void* packed, const void* packed_bwd, const float* x,
                                        const int64_t* t, int t_len, const float* d_y, void* workspace, void* bwd_workspace,
                                        const float* drop_masks, float* grad_flat, float* d_x, void* stream, dmme_bucket_fn ready, void* user) {
    DMME_REQUIRE(ready, DMME_ERR_INVALID, "unet_backward_buckets: null callback");
    return backward_impl(plan, packed, packed_bwd, x, t, t_len, d_y, workspace, bwd_workspace, drop_masks, grad_flat, d_x, stream, ready, user);
}

DMME_API int dmme_unet_plan_grad_buckets(const dmme_plan* plan, int64_t* offsets, int64_t* numels, int* bucket_of, int cap) {
    DMME_REQUIRE(plan && offsets && numels && cap > 0, DMME_ERR_INVALID, "grad_buckets: bad argument");
    if (plan->gb.empty()) {  // no clean cut for this configuration: one piece
        offsets[0] = 0;
        numels[0] = plan->ref_numel;
        if (bucket_of) bucket_of[0] = 0;
        return 1;
    }
    int n = 0;
    for (size_t b = 0; b < plan->gb.size(); ++b)
        for (const auto& r : plan->gb[b].ranges) {
            if (n < cap) {
                offsets[n] = r.first;
                numels[n] = r.second;
                if (bucket_of) bucket_of[n] = (int)b;
            }
            ++n;
        }
    return n;
}

/* Which kernels a backward of this plan launches, as "key=value" pairs: the grouped weight-gradient layers / jobs per kernel
 * size, the grouped column-sum and bias jobs, and how many DATA-gradient convolutions run on each forward kernel (by label).
 * Lets a parity test assert that a configuration really exercises the kernels it is meant to cover. */
DMME_API int dmme_unet_plan_bwd_summary(const dmme_plan* plan, char* buf, int cap) {
    DMME_REQUIRE(plan && buf && cap > 0, DMME_ERR_INVALID, "bwd_summary: bad argument");
    const dmme_plan* P = plan;
    std::string out;
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "wgrad_group3x3_layers=%d wgrad_group3x3_jobs=%d wgrad_group1x1_layers=%d wgrad_group1x1_jobs=%d colsum_group_jobs=%d bias_group_jobs=%d",
             (int)P->wg[0].layers.size(), (int)P->wg[0].jobs.size(), (int)P->wg[1].layers.size(), (int)P->wg[1].jobs.size(), (int)P->col_jobs.size(),
             (int)P->bias_jobs.size());
    out = tmp;
    std::unordered_map<std::string, int> dgrad;
    for (const Op& o : P->ops) {
        if (o.kind != OP_CONV || o.src1 < 0) continue;
        ConvArgs a{};
        fill_conv(P, o, (const char*)4096, (const float*)4096, (float*)4096, (char*)4096, nullptr, 1, a);
        ConvArgs d{};
        d.src1 = (const void*)4096;
        d.C1 = a.Cout;
        d.N = P->B;
        d.Hin = a.Hout;
        d.Win = a.Wout;
        d.up = o.stride == 2 ? 2 : 0;
        d.stride = 1;
        d.taps = o.taps;
        d.Hout = d.up ? 2 * d.Hin : d.Hin;
        d.Wout = d.up ? 2 * d.Win : d.Win;
        d.Cout = a.C1 + a.C2;
        d.w = (const void*)4096;
        d.dst = (void*)4096;
        d.x3 = P->x3;
        d.f16 = P->dtype == DMME_F16;  // (launchers without a dtype argument: conv1x1_as, the thin output conv)
        if (P->splitk_floats > 0) {
            d.splitk = (float*)4096;
            d.splitk_cap = P->splitk_floats;
        }
        char label[128] = "generic";
        if (conv1x1_pipe_supported(P->dtype, d))
            conv1x1_pipe_label(P->dtype, d, label, sizeof(label));
        else if (conv_pipe_supported(P->dtype, d))
            conv_pipe_label(P->dtype, d, label, sizeof(label));
        else if (conv_mfma_supported(P->dtype, d))
            conv_mfma_label(P->dtype, d, label, sizeof(label));
        dgrad[label] += 1;
    }
    for (const auto& kv : dgrad) {
        snprintf(tmp, sizeof(tmp), " dgrad[%s]=%d", kv.first.c_str(), kv.second);
        out += tmp;
    }
    strncpy(buf, out.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return DMME_OK;
}

}  // extern "C"
