// Network graph, workspace planner, forward and backward schedules of the segmentation engine (engine_internal.h has the overview).
#include "engine_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------------
struct Builder {
    seg_engine& e;
    int nd;
    explicit Builder(seg_engine& e_) : e(e_), nd(e_.ndim) {}

    int param(const std::string& name, std::vector<int> shape) {
        Param p; p.name = name; p.shape = shape; p.off = e.nparam; p.numel = 1;
        for (int s : shape) p.numel *= s;
        // keep every tensor 64-float aligned inside the flat buffer (vector loads, 256-B alignment)
        e.nparam += (p.numel + 63) / 64 * 64;
        e.params.push_back(p);
        return (int)e.params.size() - 1;
    }
    std::vector<int> kshape(int a, int b, int k) {
        std::vector<int> s{a, b};
        for (int i = 0; i < nd; ++i) s.push_back(k);
        return s;
    }
    int tensor(int C, int lvl, bool image = false) {
        Ten t; t.C = C; t.lvl = lvl; t.image = image;
        e.tens.push_back(t);
        return (int)e.tens.size() - 1;
    }
    // conv (+ optional GroupNorm params gw/gb: -2 => create "<gn>.weight/.bias")
    int unit(int ck, const std::string& cname, bool bias, int in0, int in1, int Cout, int lvl_out,
             const std::string& gname, int gw = -2, int gb = -2, bool has_gn = true) {
        Step s; s.type = ST_UNIT; s.ck = ck; s.in0 = in0; s.in1 = in1;
        s.Cin = e.tens[in0].C + (in1 >= 0 ? e.tens[in1].C : 0);
        s.Cout = Cout;
        if (e.tens[in0].image && e.pad_img) s.cin_par = e.in_ch;        // the parameter keeps the reference's shape [Cout][image channels][k^d]
        const int cpar = s.cin_par ? s.cin_par : s.Cin;
        const int k = (ck == CK_K3 || ck == CK_STEM3) ? 3 : (ck == CK_K2S2 || ck == CK_KT) ? 2 : 1;
        s.w = param(cname + ".weight", ck == CK_KT ? kshape(s.Cin, Cout, k) : kshape(Cout, cpar, k));
        if (bias) s.b = param(cname + ".bias", {Cout});
        if (has_gn) {
            if (gw == -2) { gw = param(gname + ".weight", {Cout}); gb = param(gname + ".bias", {Cout}); }
            s.gn_w = gw; s.gn_b = gb;
            s.mask_slot = (int)e.drop_ch.size();
            e.drop_ch.push_back(Cout);
        }
        s.raw = tensor(Cout, lvl_out);
        e.steps.push_back(s);
        return (int)e.steps.size() - 1;
    }
    int act(int ua, int ub, int res) {
        Step s; s.type = ST_ACT; s.ua = ua; s.ub = ub; s.res = res;
        const Ten& r = e.tens[e.steps[ua].raw];
        s.out = tensor(r.C, r.lvl);
        e.steps.push_back(s);
        return s.out;
    }
    int pool(int in) {
        Step s; s.type = ST_POOL; s.in = in;
        s.out = tensor(e.tens[in].C, e.tens[in].lvl + 1);
        e.steps.push_back(s);
        return s.out;
    }
    void head(int in, const std::string& cname) {
        Step s; s.type = ST_HEAD; s.in = in; s.Cin = e.tens[in].C; s.Cout = e.ncls;
        s.w = param(cname + ".weight", kshape(e.ncls, s.Cin, 1));
        s.b = param(cname + ".bias", {e.ncls});
        e.steps.push_back(s);
    }

    void build_vnet() {   // networks/VNet3d.py:102-158
        const int F = e.feat;
        const int x = tensor(e.pad_img ? 16 : e.in_ch, 0, true);
        e.image_ten = x;
        // InputTransition (VNet3d.py:25-43): parameter order conv1, conv2, bn1; ONE GroupNorm for both branches
        const int ua = unit(e.pad_img ? CK_K3 : CK_STEM3, "in_tr.conv1", true, x, -1, F, 0, "", -1, -1, false);
        const int ub = unit(e.pad_img ? CK_K1 : CK_STEM1, "in_tr.conv2", true, x, -1, F, 0, "", -1, -1, false);
        const int gw = param("in_tr.bn1.weight", {F}), gb = param("in_tr.bn1.bias", {F});
        for (int u : {ua, ub}) {
            e.steps[u].gn_w = gw; e.steps[u].gn_b = gb;
            e.steps[u].mask_slot = (int)e.drop_ch.size();
            e.drop_ch.push_back(F);
        }
        int prev = act(ua, ub, -1);
        std::vector<int> skips{prev};
        const int nconv_down[4] = {2, 3, 3, 3};
        for (int l = 1; l <= 4; ++l) {   // DownTransition (VNet3d.py:46-59)
            const int C = F << l;
            const std::string pre = "down_tr" + std::to_string(32 << (l - 1));
            const int ud = unit(CK_K2S2, pre + ".down_conv", true, prev, -1, C, l, pre + ".bn1");
            const int down = act(ud, -1, -1);
            int t = down;
            for (int i = 0; i < nconv_down[l - 1]; ++i) {
                const std::string op = pre + ".ops." + std::to_string(i);
                const int u = unit(CK_K3, op + ".conv1", true, t, -1, C, l, op + ".bn1");
                t = act(u, -1, i == nconv_down[l - 1] - 1 ? down : -1);
            }
            prev = t;
            skips.push_back(prev);
        }
        skips.pop_back();
        const int nconv_up[4] = {3, 3, 2, 1};
        for (int k = 0; k < 4; ++k) {    // UpTransition (VNet3d.py:62-80): parameter order up_conv, bn, ops, conv
            const int l = 3 - k, C = F << l;
            const std::string pre = "up_tr" + std::to_string(256 >> k);
            const int skip = skips.back(); skips.pop_back();
            const int uu = unit(CK_KT, pre + ".up_conv", true, prev, -1, C, l, pre + ".bn");
            const int gwu = e.steps[uu].gn_w, gbu = e.steps[uu].gn_b;
            const int up = act(uu, -1, -1);
            // the LUConv parameters are registered BEFORE `conv` in the reference module; keep state_dict order
            // by creating the ops' parameters first and the 1^d conv's afterwards.
            std::vector<int> opw, opb, opgw, opgb;
            for (int i = 0; i < nconv_up[k]; ++i) {
                const std::string op = pre + ".ops." + std::to_string(i);
                opw.push_back(param(op + ".conv1.weight", kshape(C, C, 3)));
                opb.push_back(param(op + ".conv1.bias", {C}));
                opgw.push_back(param(op + ".bn1.weight", {C}));
                opgb.push_back(param(op + ".bn1.bias", {C}));
            }
            const int cw = param(pre + ".conv.weight", kshape(C, 2 * C, 1));
            const int cb = param(pre + ".conv.bias", {C});
            const int uc = unit_preparam(CK_K1, cw, cb, up, skip, C, l, gwu, gbu);
            const int xcat = act(uc, -1, -1);
            int t = xcat;
            for (int i = 0; i < nconv_up[k]; ++i) {
                const int u = unit_preparam(CK_K3, opw[i], opb[i], t, -1, C, l, opgw[i], opgb[i]);
                t = act(u, -1, i == nconv_up[k] - 1 ? xcat : -1);
            }
            prev = t;
        }
        head(prev, "out_tr.conv");
    }
    int unit_preparam(int ck, int w, int b, int in0, int in1, int Cout, int lvl, int gw, int gb) {
        Step s; s.type = ST_UNIT; s.ck = ck; s.in0 = in0; s.in1 = in1;
        s.Cin = e.tens[in0].C + (in1 >= 0 ? e.tens[in1].C : 0);
        s.Cout = Cout; s.w = w; s.b = b; s.gn_w = gw; s.gn_b = gb;
        s.mask_slot = (int)e.drop_ch.size();
        e.drop_ch.push_back(Cout);
        s.raw = tensor(Cout, lvl);
        e.steps.push_back(s);
        return (int)e.steps.size() - 1;
    }

    int unet_block(const std::string& mod, const std::string& name, int in0, int in1, int C, int lvl, bool first) {
        // Unet3d.py:64-86: conv3(no bias) GN drop relu, twice
        const int u1 = unit((first && !e.pad_img) ? CK_STEM3 : CK_K3, mod + "." + name + "conv1", false, in0, in1, C, lvl, mod + "." + name + "norm1");
        const int a1 = act(u1, -1, -1);
        const int u2 = unit(CK_K3, mod + "." + name + "conv2", false, a1, -1, C, lvl, mod + "." + name + "norm2");
        return act(u2, -1, -1);
    }
    void build_unet() {   // networks/Unet3d.py:6-62
        const int F = e.feat;
        const int x = tensor(e.pad_img ? 16 : e.in_ch, 0, true);
        e.image_ten = x;
        int t = x;
        std::vector<int> enc;
        for (int l = 0; l < 4; ++l) {
            const std::string nm = "enc" + std::to_string(l + 1);
            const int en = unet_block("encoder" + std::to_string(l + 1), nm, t, -1, F << l, l, l == 0);
            enc.push_back(en);
            t = pool(en);
        }
        t = unet_block("bottleneck", "bottleneck", t, -1, F << 4, 4, false);
        for (int l = 3; l >= 0; --l) {
            const std::string up = "upconv" + std::to_string(l + 1);
            const int uu = unit(CK_KT, up, true, t, -1, F << l, l, "", -1, -1, false);
            // plain ConvTranspose: its raw output IS the activation fed to the concat
            t = unet_block("decoder" + std::to_string(l + 1), "dec" + std::to_string(l + 1), e.steps[uu].raw, enc[l], F << l, l, false);
        }
        head(t, "conv");
    }
};

Taps make_taps(int ndim, int k, int pad) {
    Taps t; t.n = 0;
    const int kd = ndim == 3 ? k : 1;
    for (int a = 0; a < kd; ++a)
        for (int b = 0; b < k; ++b)
            for (int c = 0; c < k; ++c) {
                t.d[t.n] = (int8_t)(ndim == 3 ? a - pad : 0);
                t.h[t.n] = (int8_t)(b - pad);
                t.w[t.n] = (int8_t)(c - pad);
                ++t.n;
            }
    return t;
}

// weight-gradient launch arguments of a UNIT (pointers are null until the engine is bound)
WgradArgs make_wgrad_args(const seg_engine& E, const Step& s, int draw) {
    const Ten& i0 = E.tens[s.in0];
    const Ten& ro = E.tens[s.raw];
    const int li = i0.lvl, lo = ro.lvl;
    const int T = (s.ck == CK_K3 || s.ck == CK_STEM3) ? (E.ndim == 3 ? 27 : 9)
                  : (s.ck == CK_K2S2 || s.ck == CK_KT) ? (E.ndim == 3 ? 8 : 4) : 1;
    char* ws = E.ws;
    auto P = [&](size_t off) -> const void* { return ws ? ws + off : nullptr; };
    WgradArgs w{};
    w.dw = E.g ? E.g + E.params[s.w].off : nullptr; w.N = E.N; w.sT = 1; w.sQ = T;
    if (s.ck == CK_KT) {
        // dW[ci][co][a] = sum_coarse X[m][ci] * dY[2m+a][co]
        w.dr = P(i0.off); w.P = s.Cin;
        w.x0 = draw >= 0 ? P(E.tens[draw].off) : nullptr; w.C0 = s.Cout; w.x1 = nullptr; w.C1 = 0; w.Q = s.Cout;
        w.ID = E.dim_d(lo); w.IH = E.dim_h(lo); w.IW = E.dim_w(lo);
        w.OD = E.dim_d(li); w.OH = E.dim_h(li); w.OW = E.dim_w(li);
        w.sd = E.ndim == 3 ? 2 : 1; w.sh = 2; w.sw = 2;
        w.taps = make_taps(E.ndim, 2, 0);
        w.sP = (long long)s.Cout * T;
    } else {
        w.dr = draw >= 0 ? P(E.tens[draw].off) : nullptr; w.P = s.Cout;
        w.x0 = P(i0.off); w.C0 = i0.C;
        w.x1 = s.in1 >= 0 ? P(E.tens[s.in1].off) : nullptr;
        w.C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
        w.Q = s.Cin;
        w.ID = E.dim_d(li); w.IH = E.dim_h(li); w.IW = E.dim_w(li);
        w.OD = E.dim_d(lo); w.OH = E.dim_h(lo); w.OW = E.dim_w(lo);
        const int k = (s.ck == CK_K3 || s.ck == CK_STEM3) ? 3 : s.ck == CK_K2S2 ? 2 : 1;
        const int str = s.ck == CK_K2S2 ? 2 : 1;
        w.sd = E.ndim == 3 ? str : 1; w.sh = str; w.sw = str;
        w.taps = make_taps(E.ndim, k, k == 3 ? 1 : 0);
        w.sP = (long long)(s.cin_par ? s.cin_par : s.Cin) * T;
        if (s.ck == CK_STEM3 || s.ck == CK_STEM1) { w.stem = 1; w.Q = T * s.Cin; }
        if (s.vact_unit >= 0) {          // x0 = the producer's raw output, activated on load
            const Step& pu = E.steps[s.vact_unit];
            w.x0 = P(E.tens[pu.raw].off);
            w.act_scale = (const float*)P(pu.scale); w.act_shift = (const float*)P(pu.shift);
            if (!ws) { w.act_scale = w.act_shift = (const float*)(uintptr_t)16; }      // (extents only: a non-null marker)
        }
    }
    return w;
}

// arguments of the forward launch of a generic (non-halo) conv UNIT: 2^d stride-2, 1^d on a (virtual) concat, ConvTranspose
ConvArgs make_fwd_conv_args(const seg_engine& E, const Step& s) {
    const Ten& i0 = E.tens[s.in0];
    const Ten& ro = E.tens[s.raw];
    char* ws = E.ws;
    auto P = [&](size_t off) -> char* { return ws ? ws + off : nullptr; };
    ConvArgs a{};
    a.in0 = P(i0.off); a.C0 = i0.C;
    a.in1 = s.in1 >= 0 ? P(E.tens[s.in1].off) : nullptr;
    a.C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
    a.w = P(s.wp_fwd); a.bias = (s.b >= 0 && E.p) ? E.p + E.params[s.b].off : nullptr; a.out = P(ro.off);
    a.stats = s.gn_w >= 0 ? (double*)P(s.stats) : nullptr;
    a.N = E.N; a.Cout = s.Cout;
    const int li = i0.lvl, lo = ro.lvl;
    a.ID = E.dim_d(li); a.IH = E.dim_h(li); a.IW = E.dim_w(li);
    if (s.ck == CK_KT) {
        a.scatter = 1;
        a.OD = a.ID; a.OH = a.IH; a.OW = a.IW;
        a.FD = E.dim_d(lo); a.FH = E.dim_h(lo); a.FW = E.dim_w(lo);
        a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
        a.taps = make_taps(E.ndim, 2, 0);
        a.K = s.Cin; a.Ngemm = a.taps.n * s.Cout;
    } else {
        a.scatter = 0;
        a.OD = E.dim_d(lo); a.OH = E.dim_h(lo); a.OW = E.dim_w(lo);
        const int k = s.ck == CK_K3 ? 3 : s.ck == CK_K2S2 ? 2 : 1;
        a.taps = make_taps(E.ndim, k, s.ck == CK_K3 ? 1 : 0);
        const int str = s.ck == CK_K2S2 ? 2 : 1;
        a.sd = E.ndim == 3 ? str : 1; a.sh = str; a.sw = str;
        a.K = a.taps.n * s.Cin; a.Ngemm = s.Cout;
    }
    a.Kpad = (a.K + 31) / 32 * 32;
    if (s.vact_unit >= 0) {              // in0 = the producer's raw output, activated on load
        const Step& pu = E.steps[s.vact_unit];
        a.in0 = P(E.tens[pu.raw].off);
        a.act_scale = (const float*)P(pu.scale); a.act_shift = (const float*)P(pu.shift);
        if (!ws) { a.act_scale = a.act_shift = (const float*)(uintptr_t)16; }          // (extents only: a non-null marker)
    }
    return a;
}

// arguments of the fused input block behind ACT step `s` (pointers valid once the engine is bound)
seg_stemx_args stemx_args(const seg_engine& E, const Step& s) {
    const Step& ua = E.steps[s.ua];
    seg_stemx_args x{};
    x.img = E.ws + E.tens[ua.in0].off;
    x.w3 = E.ws + ua.wp_fwd; x.bias3 = ua.b >= 0 ? E.p + E.params[ua.b].off : nullptr;
    x.stats3 = (double*)(E.ws + ua.stats); x.scale3 = (float*)(E.ws + ua.scale); x.shift3 = (float*)(E.ws + ua.shift);
    x.Q3 = (double*)(E.ws + ua.Q); x.coef3 = (float*)(E.ws + ua.coef);
    if (s.ub >= 0) {
        const Step& ub = E.steps[s.ub];
        x.w1 = E.ws + ub.wp_fwd; x.bias1 = ub.b >= 0 ? E.p + E.params[ub.b].off : nullptr;
        x.stats1 = (double*)(E.ws + ub.stats); x.scale1 = (float*)(E.ws + ub.scale); x.shift1 = (float*)(E.ws + ub.shift);
        x.Q1 = (double*)(E.ws + ub.Q); x.coef1 = (float*)(E.ws + ub.coef);
    }
    x.out = E.ws + E.tens[s.out].off;
    x.partial = (float*)(E.ws + E.off_partial_stemx);
    x.N = E.N; x.D = E.dim_d(0); x.H = E.dim_h(0); x.W = E.dim_w(0); x.Cimg = E.tens[ua.in0].C;
    return x;
}

// both data-gradients of a 1^d conv on a (virtual) concat as ONE streaming launch over d(raw) (their packed weights lie back to back: one [C0 + C1][Kpad]
// matrix); `draw`, g0, g1: tensor ids or -1 (extents only).  Returns false where the launch does not apply.
bool make_dual_dgrad_args(const seg_engine& E, const Step& s, int draw, int g0, int g1, ConvArgs& b) {
    if (s.ck != CK_K1 || s.in1 < 0 || E.tens[s.in0].image) return false;
    const int lo = E.tens[s.raw].lvl, C0 = E.tens[s.in0].C, C1 = E.tens[s.in1].C;
    char* ws = E.ws;
    auto P = [&](size_t off) -> char* { return ws ? ws + off : nullptr; };
    b = ConvArgs{};
    b.in0 = draw >= 0 ? P(E.tens[draw].off) : nullptr; b.C0 = s.Cout; b.N = E.N;
    b.scatter = 0;
    b.ID = b.OD = E.dim_d(lo); b.IH = b.OH = E.dim_h(lo); b.IW = b.OW = E.dim_w(lo);
    b.sd = b.sh = b.sw = 1;
    b.taps = make_taps(E.ndim, 1, 0);
    b.K = s.Cout; b.Kpad = (b.K + 31) / 32 * 32;
    b.w = P(s.wp_dg0); b.out = g0 >= 0 ? P(E.tens[g0].off) : nullptr; b.out1 = g1 >= 0 ? P(E.tens[g1].off) : (void*)(uintptr_t)16;
    b.Cout0 = C0; b.Cout = b.Ngemm = C0 + C1;
    return s.wp_dg1 == s.wp_dg0 + (size_t)C0 * b.Kpad * E.esz() && conv_uses_stream_kernel(b);
}

// ------------------------------------------------------------------------------------------------
// planning: workspace layout + forward / backward schedules
// ------------------------------------------------------------------------------------------------
struct Planner {
    seg_engine& e;
    size_t cur = 0;
    explicit Planner(seg_engine& e_) : e(e_) {}
    size_t alloc(size_t bytes) { size_t o = cur; cur = align_up(cur + bytes); return o; }
    size_t ten_bytes(const Ten& t) const { return (size_t)e.N * e.vol(t.lvl) * t.C * e.esz(); }
    int new_grad(int like) {
        Ten t; t.C = e.tens[like].C; t.lvl = e.tens[like].lvl;
        t.off = alloc(ten_bytes(t));
        e.tens.push_back(t);
        return (int)e.tens.size() - 1;
    }
    template <class T = void> T* P(size_t off) const { return (T*)(e.ws + off); }

    int ntaps(int ck) const {
        const int k = (ck == CK_K3 || ck == CK_STEM3) ? 3 : (ck == CK_K2S2 || ck == CK_KT) ? 2 : 1;
        return e.ndim == 3 ? k * k * k : k * k;
    }
    bool pack_bwd = false;     // the descriptors added while set feed the backward pass only (data-gradient layouts)
    void add_pack(size_t dst, long long src_off, int R1, int R2, int T, int Cc, long long s1, long long s2, long long sT, long long sC, int flip,
                  int frag = 0, int csrc = 0) {
        PackDesc d;
        d.frag = frag;
        d.csrc = csrc;
        d.src = (const float*)(uintptr_t)src_off;   // offsets; resolved in seg_bind
        d.dst = (void*)(uintptr_t)dst;
        d.R1 = R1; d.R2 = R2; d.T = T; d.Cc = Cc;
        d.Kpad = frag == 3 ? 480 : (T * Cc + 31) / 32 * 32;      // frag 3: 15 steps of two 16-channel taps (conv3x16r_kernel)
        d.s1 = s1; d.s2 = s2; d.sT = sT; d.sC = sC; d.flipT = flip;
        e.packdescs.push_back(d);
        e.pack_is_bwd.push_back(pack_bwd ? 1 : 0);
        const long long tot = (long long)R1 * R2 * d.Kpad;
        if (tot > e.pack_max) e.pack_max = tot;
    }
    size_t alloc_pack(int rows, int K, int frag = 0) { return alloc((size_t)rows * (frag == 3 ? 480 : (K + 31) / 32 * 32) * e.esz()); }

    void plan() {
        seg_engine& E = e;
        const int N = E.N, dt = E.dtype;
        E.fwd_ops.clear(); E.bwd_ops.clear(); E.bwd_writes.clear(); E.packdescs.clear(); E.pack_is_bwd.clear(); E.pack_max = 0;
        // drop gradient tensors of a previous plan
        size_t nfw = 0;
        for (auto& s : E.steps) {
            nfw = std::max<size_t>(nfw, std::max(s.raw, s.out) + 1);
            s.draw = -1;
        }
        E.tens.resize(std::max<size_t>(nfw, (size_t)E.image_ten + 1));
        for (auto& t : E.tens) t.grads.clear();

        // ---- fused input block: an ACT whose unit(s) are image stems (3^d [+ 1^d]) without a residual
        for (auto& st_ : E.steps) st_.fused_stem = false;
        if (E.use_stemx && E.feat == 16 && (long long)E.vol(0) * 16 * 4 < (1ll << 31))
            for (auto& st_ : E.steps)
                if (st_.type == ST_ACT && st_.res < 0 && E.steps[st_.ua].ck == CK_STEM3 && E.steps[st_.ua].gn_w >= 0 &&
                    (st_.ub < 0 || (E.steps[st_.ub].ck == CK_STEM1 && E.steps[st_.ub].gn_w >= 0))) {
                    E.steps[st_.ua].fused_stem = true;
                    if (st_.ub >= 0) E.steps[st_.ub].fused_stem = true;
                }
        // ---- activations that are never written: the output of a single-branch ACT step without residual whose ONLY reader is the first source of a 1^d conv on a
        // (virtual) concat - the VNet up-conv -> concat -> conv chain, networks/VNet3d.py:72-77 - on tensors large enough for the two passes over it to cost
        // bandwidth (>= 16 MB: the 96^3 and 48^3 decoder levels of the benchmark; below, the launches are latency and the generic conv kernel applies).  The
        // reader's forward launch (streaming conv kernel) and its weight gradient (direct kernel) take the producer's raw output and apply scale / shift / ReLU on
        // load; the unit's gradient flow is unchanged.  16-bit run dtypes.
        for (auto& st_ : E.steps) { st_.vact = false; st_.vact_unit = -1; }
        if (E.use_vact && dt != DT_F32)
            for (size_t ai = 0; ai < E.steps.size(); ++ai) {
                Step& A = E.steps[ai];
                if (A.type != ST_ACT || A.ub >= 0 || A.res >= 0 || E.steps[A.ua].fused_stem || E.steps[A.ua].gn_w < 0) continue;
                if (E.use_vact < 2 && (double)ten_bytes(E.tens[A.out]) < 16e6) continue;
                int readers = 0, ci = -1;
                for (size_t k = 0; k < E.steps.size(); ++k) {
                    const Step& c = E.steps[k];
                    if (c.type == ST_UNIT) { if (c.in0 == A.out) { ++readers; ci = (int)k; } if (c.in1 == A.out) readers += 2; }
                    else if (c.type == ST_ACT) { if (c.res == A.out) readers += 2; }
                    else if (c.in == A.out) readers += 2;
                }
                if (readers != 1) continue;
                Step& c = E.steps[ci];
                if (c.ck != CK_K1 || E.tens[c.in0].image) continue;
                c.vact_unit = A.ua;
                char* keep = E.ws; E.ws = nullptr;
                const bool ok = conv_uses_stream_kernel(make_fwd_conv_args(E, c)) && wgrad_act_supported(make_wgrad_args(E, c, -1));
                E.ws = keep;
                if (ok) A.vact = true; else c.vact_unit = -1;
            }
        // ---- the 1^d head evaluated by the activation pass that writes its input (16 channels, <= 4 classes)
        for (auto& st_ : E.steps) st_.head_fused = false;
        if (E.use_head_fuse)
            for (auto& hs : E.steps) {
                if (hs.type != ST_HEAD) continue;
                for (auto& A : E.steps) {
                    if (A.type != ST_ACT || A.out != hs.in || A.vact || A.ub >= 0 || E.steps[A.ua].fused_stem || E.steps[A.ua].gn_w < 0) continue;
                    const Step& ua_ = E.steps[A.ua];
                    if (gn_bwd_group_eligible(ua_.Cout, E.vol(E.tens[ua_.raw].lvl), (int)E.esz())) continue;      // (one-launch small-tensor pass)
                    if (!gn_act_head_supported(E.tens[A.out].C, hs.Cout, false) || E.tens[A.out].lvl != 0) continue;
                    A.head_fused = true; hs.head_fused = true;
                }
            }
        // ---- statistics finalize folded into the elementwise consumer (not for the fused input block / one-launch small tensors)
        for (auto& st_ : E.steps) st_.fold_fin = false;
        if (E.use_fold)
            for (auto& st_ : E.steps) {
                if (st_.type != ST_ACT || st_.vact || E.steps[st_.ua].fused_stem || E.steps[st_.ua].gn_w < 0) continue;
                const Step& ua_ = E.steps[st_.ua];
                if (st_.ub < 0 && gn_bwd_group_eligible(ua_.Cout, E.vol(E.tens[ua_.raw].lvl), (int)E.esz())) continue;
                if (ua_.Cout > 256) continue;
                E.steps[st_.ua].fold_fin = true;
                if (st_.ub >= 0) E.steps[st_.ub].fold_fin = true;
            }
        // ---- small persistent regions
        E.off_step = alloc(256);
        E.off_masks = alloc((size_t)E.drop_ch.size() * N * E.ld_mask() * 4);
        // forward tensors
        for (auto& t : E.tens) t.off = alloc(ten_bytes(t));
        // statistics (fp64) contiguous so one memset clears them; same for Q
        const size_t s0 = cur;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) s.stats = alloc((size_t)STAT_REP * N * s.Cout * 2 * 8);
        E.off_stats = s0; E.stats_bytes = cur - s0;
        const size_t q0 = cur;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) s.Q = alloc((size_t)STAT_REP * N * s.Cout * 2 * 8);
        E.off_Q = q0; E.Q_bytes = cur - q0;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) {
                s.scale = alloc((size_t)N * s.Cout * 4);
                s.shift = alloc((size_t)N * s.Cout * 4);
                s.mean = alloc((size_t)N * GN_GROUPS * 4);
                s.rstd = alloc((size_t)N * GN_GROUPS * 4);
                s.coef = alloc((size_t)N * s.Cout * 3 * 4);
            }
        // ---- packed weights
        for (auto& s : E.steps) {
            if (s.type != ST_UNIT) continue;
            const int T = ntaps(s.ck), Ci = s.Cin, Co = s.Cout;
            const long long woff = E.params[s.w].off;
            const int C0 = E.tens[s.in0].C, C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
            switch (s.ck) {
                case CK_K3: case CK_K1: case CK_K2S2:
                    s.x_fwd = s.x_dg0 = s.x_dg1 = -1;
                    if (s.ck == CK_K3 && E.use_conv3x) {
                        // register-blocked halo kernel (conv3x.hip) wherever the shape allows: fragment-major weights
                        const int l = E.tens[s.raw].lvl, d_ = E.dim_d(l), h_ = E.dim_h(l), w_ = E.dim_w(l);
                        if (conv3x_supported(dt, E.ndim, N, d_, h_, w_, Ci, Co, C0, C1 > 0)) s.x_fwd = conv3x_pick(E.ndim, N, d_, h_, w_, Ci, Co, C1 > 0);
                        if (!E.tens[s.in0].image && conv3x_supported(dt, E.ndim, N, d_, h_, w_, Co, C0, 0, false))
                            s.x_dg0 = conv3x_pick(E.ndim, N, d_, h_, w_, Co, C0);
                        if (C1 && conv3x_supported(dt, E.ndim, N, d_, h_, w_, Co, C1, 0, false)) s.x_dg1 = conv3x_pick(E.ndim, N, d_, h_, w_, Co, C1);
                    }
                    s.wp_fwd = alloc_pack(Co, T * Ci, conv3x_cfg_frag(s.x_fwd));
                    add_pack(s.wp_fwd, woff, Co, 1, T, Ci, (long long)(s.cin_par ? s.cin_par : Ci) * T, 0, 1, T, 0, s.x_fwd >= 0 ? conv3x_cfg_frag(s.x_fwd) : 0,
                             s.cin_par);                         // image convs on a zero-padded image tensor: the parameter has cin_par channels
                    if (s.ck == CK_K2S2) {       // data-gradient = scatter GEMM, rows (a, ci), K = Cout
                        s.wp_dg0 = alloc_pack(T * Ci, Co);
                        pack_bwd = true;
                        add_pack(s.wp_dg0, woff, T, Ci, 1, Co, 1, T, 0, (long long)Ci * T, 0);
                        pack_bwd = false;
                    } else {                     // data-gradient = gather conv with flipped taps, rows ci, k = (tap, co)
                        if (!E.tens[s.in0].image) {
                            s.wp_dg0 = alloc_pack(C0, T * Co, conv3x_cfg_frag(s.x_dg0));
                            pack_bwd = true;
                            add_pack(s.wp_dg0, woff, C0, 1, T, Co, T, 0, 1, (long long)Ci * T, 1, s.x_dg0 >= 0 ? conv3x_cfg_frag(s.x_dg0) : 0);
                            pack_bwd = false;
                        }
                        if (C1) {
                            s.wp_dg1 = alloc_pack(C1, T * Co, conv3x_cfg_frag(s.x_dg1));
                            pack_bwd = true;
                            add_pack(s.wp_dg1, woff + (long long)C0 * T, C1, 1, T, Co, T, 0, 1, (long long)Ci * T, 1, s.x_dg1 >= 0 ? conv3x_cfg_frag(s.x_dg1) : 0);
                            pack_bwd = false;
                        }
                    }
                    break;
                case CK_KT:                      // forward = scatter GEMM rows (a, co), K = Cin
                    s.wp_fwd = alloc_pack(T * Co, Ci);
                    add_pack(s.wp_fwd, woff, T, Co, 1, Ci, 1, T, 0, (long long)Co * T, 0);
                    s.wp_dg0 = alloc_pack(Ci, T * Co);   // data-gradient = gather stride 2, rows ci, k = (a, co)
                    pack_bwd = true;
                    add_pack(s.wp_dg0, woff, Ci, 1, T, Co, (long long)Co * T, 0, 1, T, 0);
                    pack_bwd = false;
                    break;
                default:                         // image stems: [Cout][32] with k = tap*Cimg + ci (1^d stem: k = ci)
                    s.wp_fwd = alloc_pack(Co, T * Ci);
                    add_pack(s.wp_fwd, woff, Co, 1, T, Ci, (long long)Ci * T, 0, 1, T, 0);
                    break;
            }
        }
        {   // forward layouts first, backward-only layouts behind them: the second range is packed on the weight-gradient stream
            std::vector<PackDesc> fw, bw;
            for (size_t i = 0; i < E.packdescs.size(); ++i) (E.pack_is_bwd[i] ? bw : fw).push_back(E.packdescs[i]);
            E.npack_fwd = (int)fw.size();
            E.packdescs = fw;
            E.packdescs.insert(E.packdescs.end(), bw.begin(), bw.end());
            E.pack_is_bwd.assign(E.packdescs.size(), 0);
            for (size_t i = fw.size(); i < E.packdescs.size(); ++i) E.pack_is_bwd[i] = 1;
        }
        E.off_packdesc = alloc(E.packdescs.size() * sizeof(PackDesc));
        // ---- 1^d convs on a concat: one data-gradient launch for both sources; where the first source is a never-written activation (vact) of equal width, that
        // launch also carries the GroupNorm-backward sums of the activation's unit
        for (auto& st_ : E.steps) { st_.dual_dg = false; st_.rq_fused = false; }
        for (auto& c : E.steps) {
            if (c.type != ST_UNIT) continue;
            ConvArgs b;
            char* keep = E.ws; E.ws = nullptr;
            c.dual_dg = make_dual_dgrad_args(E, c, -1, -1, -1, b);
            E.ws = keep;
            if (!c.dual_dg || c.vact_unit < 0 || !E.use_rq_fuse || E.tens[c.in0].C != E.tens[c.in1].C) continue;
            for (auto& A : E.steps)
                if (A.type == ST_ACT && A.vact && A.ua == c.vact_unit) {
                    const Step& pu = E.steps[A.ua];
                    GnBwdArgs probe{}; probe.ndy = 1; probe.C = pu.Cout; probe.V = E.vol(E.tens[pu.raw].lvl); probe.N = N;
                    if (E.use_coop && gn_bwd_coop_eligible(probe, (int)E.esz())) continue;
                    if (gn_bwd_group_eligible(pu.Cout, probe.V, (int)E.esz())) continue;
                    A.rq_fused = true;
                }
        }
        // partial-tile buffer of the halo weight-gradient kernel (largest K3 layer)
        size_t pmax = 0;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT) {
                if (s.ck == CK_K3) {
                    const int l = E.tens[s.raw].lvl;
                    pmax = std::max(pmax, wgrad3_partial_bytes(E.ndim, N, E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cout, s.Cin));
                } else if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                    pmax = std::max(pmax, stem_wgrad_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), s.Cout));
                } else {
                    char* keep = E.ws; E.ws = nullptr;
                    pmax = std::max(pmax, wgrad_partial_bytes(make_wgrad_args(E, s, -1)));
                    E.ws = keep;
                }
            }
        E.off_partial = alloc(pmax);
        {   // halo weight gradients keep one partial-tile slot per layer until the reduce of their level visit (seg_engine::w3_slot)
            size_t p3 = 0;
            for (auto& s : E.steps)
                if (s.type == ST_UNIT && s.ck == CK_K3) {
                    const int l = E.tens[s.raw].lvl;
                    p3 = std::max(p3, wgrad3_partial_bytes(E.ndim, N, E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cout, s.Cin));
                }
            E.partial3_stride = align_up(p3);
            E.off_partial3 = alloc(E.partial3_stride * W3_BATCH);
        }
        E.off_partial_stemx = alloc(stemx_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), E.in_ch));
        E.off_partial_stem1 = alloc(stem_wgrad_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), 16 * ((E.feat + 15) / 16)));

        // ------------------------------------------------------------------ forward schedule
        E.fwd_ops.push_back([this_ = &E](hipStream_t st) {
            seg_engine& E = *this_;
            // the backward sums (Q) sit right behind the forward statistics: ONE fill clears both (a fill is a ~6 us launch on the main
            // stream); a backward pass that does not follow a forward pass directly clears Q itself
            const Ten& x = E.tens[E.image_ten];
            const size_t fill = E.stats_bytes + (E.off_Q == E.off_stats + E.stats_bytes ? E.Q_bytes : 0);
            const int pi = E.prof_begin(st, SEG_K_MISC, (double)fill + (double)E.N * E.vol(0) * (4.0 * E.in_ch + (double)x.C * E.esz()), 0.0);
            (void)hipMemsetAsync(E.ws + E.off_stats, 0, fill, st);
            E.q_clean = E.off_Q == E.off_stats + E.stats_bytes;
            launch_ingest(E.cur_x, E.ws + x.off, E.N, x.C, E.vol(0), E.dtype, st, E.in_ch, E.ride_on ? E.ride_ingest : StepRider{});
            E.prof_end(st, pi);
        });
        for (size_t si = 0; si < E.steps.size(); ++si) {
            Step& s = E.steps[si];
            if (s.type == ST_UNIT) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    if (s.fused_stem) return;              // evaluated by the fused input block of its ACT step
                    const Ten& i0 = E.tens[s.in0];
                    const Ten& ro = E.tens[s.raw];
                    double* stats = s.gn_w >= 0 ? (double*)(E.ws + s.stats) : nullptr;
                    const float* bias = s.b >= 0 ? E.p + E.params[s.b].off : nullptr;
                    if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                        const int T = s.ck == CK_STEM3 ? (E.ndim == 3 ? 27 : 9) : 1;
                        const int pi = E.prof_begin(st, SEG_K_STEM, E.tbytes(s.in0) + E.tbytes(s.raw), 2.0 * E.N * E.vol(0) * T * i0.C * s.Cout);
                        launch_stem_fwd(E.ws + i0.off, E.ws + s.wp_fwd, bias, E.ws + ro.off, stats, E.N, E.dim_d(0), E.dim_h(0), E.dim_w(0),
                                        i0.C, s.Cout, s.ck == CK_STEM1, E.ndim, E.dtype, st);
                        E.prof_end(st, pi);
                    } else if (s.ck == CK_K3) {
                        const int l = ro.lvl;
                        const int pi = E.prof_begin(st, conv3_class(E.dim_w(l), s.Cin), E.tbytes(s.in0) + E.tbytes(s.raw),
                                                    2.0 * E.N * E.vol(l) * (E.ndim == 3 ? 27 : 9) * s.Cin * s.Cout);
                        // replicas this producer spreads the statistics over (read back by the folded finalize of the consumers)
                        E.steps[si].stat_rep = (s.x_fwd >= 0 && E.use_fold) ? stat_rep_for(E.vol(l)) : STAT_REP;
                        if (s.x_fwd >= 0)
                            launch_conv3x(s.x_fwd, E.ws + i0.off, s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C, E.ws + s.wp_fwd, bias,
                                          E.ws + ro.off, stats, E.N, E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cin, s.Cout, E.ndim, E.dtype, st,
                                          s.stat_rep);
                        else
                        launch_conv3(E.ws + i0.off, E.ws + s.wp_fwd, bias, E.ws + ro.off, stats, E.N, E.dim_d(l), E.dim_h(l), E.dim_w(l),
                                     s.Cin, s.Cout, E.ndim, E.dtype, st, s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C);
                        E.prof_end(st, pi);
                    } else {
                        const ConvArgs a = make_fwd_conv_args(E, s);
                        const int li = i0.lvl, lo = ro.lvl;
                        const int pi = E.prof_begin(st, SEG_K_CONV_GENERIC,
                                                    E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0) + E.tbytes(s.raw),
                                                    2.0 * E.N * E.vol(s.ck == CK_KT ? li : lo) * (double)a.K * a.Ngemm);
                        E.steps[si].stat_rep = (E.use_fold && !conv_uses_stream_kernel(a)) ? stat_rep_for(E.vol(lo)) : STAT_REP;
                        launch_conv_igemm(a, E.dtype, st, E.steps[si].stat_rep);
                        E.prof_end(st, pi);
                    }
                    if (s.gn_w >= 0 && !s.fold_fin && !gn_bwd_group_eligible(s.Cout, E.vol(ro.lvl), (int)E.esz())) {
                        GnFinArgs f{};
                        f.stats = stats; f.gamma = E.p + E.params[s.gn_w].off; f.beta = E.p + E.params[s.gn_b].off;
                        f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                 : E.mask_base(s.mask_slot);
                        f.mask_ld = E.ld_mask();
                        f.scale = (float*)(E.ws + s.scale); f.shift = (float*)(E.ws + s.shift);
                        f.mean = (float*)(E.ws + s.mean); f.rstd = (float*)(E.ws + s.rstd);
                        f.N = E.N; f.C = s.Cout; f.V = E.vol(ro.lvl); f.eps = 1e-5f;
                        launch_gn_finalize(f, st);
                    }
                });
            } else if (s.type == ST_ACT) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Step& ua = E.steps[s.ua];
                    if (s.vact) return;                    // applied by the reader of the tensor on load (its unit's op launched the statistics finalize)
                    if (ua.fused_stem) {
                        // fused input block: statistics of both branches from the image, finalize, then recompute + normalise + add
                        seg_stemx_args x = stemx_args(E, s);
                        const int pi = E.prof_begin(st, SEG_K_STEM, E.tbytes(ua.in0) * 2 + E.tbytes(s.out), 0.0);
                        launch_stemx(x, 0, E.ndim, E.dtype, nullptr, nullptr, st);
                        GnFinArgs fin[2];
                        int nfin = 0;
                        for (int ui : {s.ua, s.ub}) {
                            if (ui < 0) continue;
                            const Step& u = E.steps[ui];
                            GnFinArgs f{};
                            f.stats = (double*)(E.ws + u.stats); f.gamma = E.p + E.params[u.gn_w].off; f.beta = E.p + E.params[u.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + u.scale); f.shift = (float*)(E.ws + u.shift);
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(0); f.eps = 1e-5f;
                            fin[nfin++] = f;
                        }
                        launch_gn_finalize(fin[0], st, nfin > 1 ? &fin[1] : nullptr);      // both branches: one launch
                        launch_stemx(x, 1, E.ndim, E.dtype, nullptr, nullptr, st);
                        E.prof_end(st, pi);
                        return;
                    }
                    {
                        const Ten& ro = E.tens[ua.raw];
                        if (s.ub < 0 && gn_bwd_group_eligible(ua.Cout, E.vol(ro.lvl), (int)E.esz())) {
                            // small L2-resident tensor: statistics finalize + activation in one launch
                            GnFinArgs f{};
                            f.stats = (double*)(E.ws + ua.stats);
                            f.gamma = E.p + E.params[ua.gn_w].off; f.beta = E.p + E.params[ua.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(ua.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + ua.scale); f.shift = (float*)(E.ws + ua.shift);
                            f.mean = (float*)(E.ws + ua.mean); f.rstd = (float*)(E.ws + ua.rstd);
                            f.N = E.N; f.C = ua.Cout; f.V = E.vol(ro.lvl); f.eps = 1e-5f; f.rep = ua.stat_rep;
                            const int pi = E.prof_begin(st, SEG_K_GN_GROUP, E.tbytes(s.out) * (2 + (s.res >= 0)), 0.0);
                            launch_gn_fwd_group(f, E.ws + ro.off, s.res >= 0 ? E.ws + E.tens[s.res].off : nullptr, E.ws + E.tens[s.out].off,
                                                E.dtype, st);
                            E.prof_end(st, pi);
                            return;
                        }
                    }
                    ActArgs a{};
                    a.r1 = E.ws + E.tens[ua.raw].off; a.scale1 = (float*)(E.ws + ua.scale); a.shift1 = (float*)(E.ws + ua.shift);
                    if (s.ub >= 0) {
                        const Step& ub = E.steps[s.ub];
                        a.r2 = E.ws + E.tens[ub.raw].off; a.scale2 = (float*)(E.ws + ub.scale); a.shift2 = (float*)(E.ws + ub.shift);
                    }
                    a.res = s.res >= 0 ? E.ws + E.tens[s.res].off : nullptr;
                    a.out = E.ws + E.tens[s.out].off;
                    a.N = E.N; a.C = E.tens[s.out].C; a.V = E.vol(E.tens[s.out].lvl);
                    if (ua.fold_fin) {
                        auto fin = [&E](const Step& u, GnFinArgs& f) {
                            f = GnFinArgs{};
                            f.stats = (double*)(E.ws + u.stats); f.gamma = E.p + E.params[u.gn_w].off; f.beta = E.p + E.params[u.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + u.scale); f.shift = (float*)(E.ws + u.shift);
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(E.tens[u.raw].lvl); f.eps = 1e-5f; f.rep = u.stat_rep;
                        };
                        a.fold = 1;
                        fin(ua, a.fin1);
                        if (s.ub >= 0) fin(E.steps[s.ub], a.fin2);
                    }
                    if (s.head_fused) {
                        const Step& hs = E.steps[E.head_step];
                        a.head_w = E.p + E.params[hs.w].off; a.head_b = E.p + E.params[hs.b].off;
                        a.logits = E.cur_logits; a.probs = E.cur_probs; a.head_C = hs.Cout;
                        if (E.ride_on && E.ride_zero) { a.zero_ptr = E.ride_zero; a.zero_n = E.ride_zero_n; E.head_zeroed = true; }
                    }
                    const int pi = E.prof_begin(st, SEG_K_GN_ACT, E.tbytes(s.out) * (2 + (s.ub >= 0) + (s.res >= 0)), 0.0);
                    launch_gn_act(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            } else if (s.type == ST_POOL) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& ti = E.tens[s.in];
                    PoolArgs a{};
                    a.in = E.ws + ti.off; a.out = E.ws + E.tens[s.out].off;
                    a.N = E.N; a.D = E.dim_d(ti.lvl); a.H = E.dim_h(ti.lvl); a.W = E.dim_w(ti.lvl); a.C = ti.C;
                    a.pd = E.ndim == 3 ? 2 : 1; a.ph = 2; a.pw = 2;
                    launch_maxpool_fwd(a, E.dtype, st);
                });
            } else {   // HEAD
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    if (s.head_fused) return;              // evaluated by the activation pass that wrote its input
                    HeadArgs a;
                    a.in = E.ws + E.tens[s.in].off; a.w = E.p + E.params[s.w].off; a.bias = E.p + E.params[s.b].off;
                    a.logits = E.cur_logits; a.probs = E.cur_probs;
                    a.N = E.N; a.V = (int)E.vol(0); a.Cin = s.Cin; a.C = s.Cout;
                    if (E.ride_on && E.ride_zero) { a.zero_ptr = E.ride_zero; a.zero_n = E.ride_zero_n; E.head_zeroed = true; }
                    const int pi = E.prof_begin(st, SEG_K_HEAD, E.tbytes(s.in) + 2.0 * 4.0 * E.N * E.vol(0) * s.Cout, 0.0);
                    launch_head_fwd(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            }
        }

        // ------------------------------------------------------------------ backward schedule
        E.bwd_writes.push_back({});
        E.bwd_ops.push_back([this_ = &E](hipStream_t st) {
            seg_engine& E = *this_;
            if (!E.q_clean) (void)hipMemsetAsync(E.ws + E.off_Q, 0, E.Q_bytes, st);
            E.q_clean = false;
        });
        for (int si = (int)E.steps.size() - 1; si >= 0; --si) {
            Step& s = E.steps[si];
            if (s.type == ST_HEAD) {
                const int gin = new_grad(s.in);
                E.tens[gin].virt = E.use_vhead;
                E.head_din_needed = !E.use_vhead;
                E.head_step = si;
                E.tens[s.in].grads.push_back(gin);
                E.bwd_writes.push_back({s.w, s.b});
                E.bwd_ops.push_back([this_ = &E, si, gin](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    HeadBwdArgs a;
                    a.in = E.ws + E.tens[s.in].off; a.w = E.p + E.params[s.w].off; a.dlogits = E.cur_dlogits;
                    // rank-K gradient: its readers (GroupNorm-backward passes) rebuild it from dlogits unless one of them cannot
                    a.din = E.head_din_needed ? E.ws + E.tens[gin].off : nullptr;
                    a.dw = E.g + E.params[s.w].off; a.db = E.g + E.params[s.b].off;
                    a.N = E.N; a.V = (int)E.vol(0); a.Cin = s.Cin; a.C = s.Cout;
                    const int pi = E.prof_begin(st, SEG_K_HEAD, E.tbytes(s.in) * (a.din ? 2.0 : 1.0) + 4.0 * E.N * E.vol(0) * s.Cout, 0.0);
                    launch_head_bwd(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            } else if (s.type == ST_POOL) {
                std::vector<int> gl = E.tens[s.out].grads;
                if (gl.size() != 1) { g_err = "internal: pool output needs exactly one gradient"; return; }
                if (E.tens[gl[0]].virt) E.head_din_needed = true;
                const int gin = new_grad(s.in);
                E.tens[s.in].grads.push_back(gin);
                const int gout = gl[0];
                E.bwd_writes.push_back({});
                E.bwd_ops.push_back([this_ = &E, si, gin, gout](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& ti = E.tens[s.in];
                    PoolArgs a{};
                    a.in = E.ws + ti.off; a.dout = E.ws + E.tens[gout].off; a.din = E.ws + E.tens[gin].off;
                    a.N = E.N; a.D = E.dim_d(ti.lvl); a.H = E.dim_h(ti.lvl); a.W = E.dim_w(ti.lvl); a.C = ti.C;
                    a.pd = E.ndim == 3 ? 2 : 1; a.ph = 2; a.pw = 2;
                    launch_maxpool_bwd(a, E.dtype, st);
                });
            } else if (s.type == ST_ACT) {
                std::vector<int> gl = E.tens[s.out].grads;
                if (gl.empty() || gl.size() > 3) { g_err = "internal: unsupported gradient fan-in"; return; }
                if (s.res >= 0) for (int gi : gl) E.tens[s.res].grads.push_back(gi);
                {
                    // the fused input block, the dual-branch and the one-launch small-tensor passes read real tensors only
                    const Step& ua_ = E.steps[s.ua];
                    const bool generic = !ua_.fused_stem && s.ub < 0 &&
                                         !gn_bwd_group_eligible(E.tens[ua_.raw].C, E.vol(E.tens[ua_.raw].lvl), (int)E.esz());
                    for (int gi : gl) if (E.tens[gi].virt && !generic) E.head_din_needed = true;
                }
                // per-branch argument builders (shared by the single- and the dual-branch op)
                auto fill = [](seg_engine& E, int ui, const std::vector<int>& gl, GnBwdArgs& a, GnBwdFinArgs& f) {
                    const Step& u = E.steps[ui];
                    const Ten& r = E.tens[u.raw];
                    a = GnBwdArgs{};
                    a.ndy = 0;
                    for (int gi : gl) {
                        if (E.tens[gi].virt && !E.head_din_needed) {
                            const Step& hs = E.steps[E.head_step];
                            a.vdl = E.cur_dlogits; a.vw = E.p + E.params[hs.w].off; a.vK = hs.Cout;
                        } else a.dy[a.ndy++] = E.ws + E.tens[gi].off;
                    }
                    a.r = E.ws + r.off;
                    a.scale = (float*)(E.ws + u.scale); a.shift = (float*)(E.ws + u.shift);
                    a.Q = (double*)(E.ws + u.Q); a.coef = (float*)(E.ws + u.coef);
                    a.dr = E.ws + E.tens[u.draw].off;
                    a.N = E.N; a.C = r.C; a.V = E.vol(r.lvl);
                    f = GnBwdFinArgs{};
                    f.Q = a.Q; f.stats = (double*)(E.ws + u.stats);
                    f.gamma = E.p + E.params[u.gn_w].off;
                    f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                             : E.mask_base(u.mask_slot);
                    f.mask_ld = E.ld_mask();
                    f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                    f.dgamma = E.g + E.params[u.gn_w].off; f.dbeta = E.g + E.params[u.gn_b].off;
                    f.dbias = u.b >= 0 ? E.g + E.params[u.b].off : nullptr;
                    f.coef = (float*)(E.ws + u.coef);
                    f.N = E.N; f.C = r.C; f.V = a.V;
                    a.rep_q = f.rep_q = E.use_fold ? stat_rep_for(a.V) : 0;
                    f.rep_s = u.stat_rep;
                };
                if (E.steps[s.ua].fused_stem) {
                    // fused input block: reduce (recomputing r), finalize per branch, then d(raw) in registers -> stem weight gradients
                    std::vector<int> wr;
                    for (int ui : {s.ua, s.ub})
                        if (ui >= 0) { const Step& u = E.steps[ui]; wr.push_back(u.gn_w); wr.push_back(u.gn_b); wr.push_back(u.b); wr.push_back(u.w); }
                    E.bwd_writes.push_back(wr);
                    E.bwd_ops.push_back([this_ = &E, si, gl](hipStream_t st) {
                        seg_engine& E = *this_;
                        const Step& s = E.steps[si];
                        seg_stemx_args x = stemx_args(E, s);
                        x.ndy = (int)gl.size();
                        for (int i = 0; i < x.ndy; ++i) x.dy[i] = E.ws + E.tens[gl[i]].off;
                        E.flush_side(st);
                        const double tb = E.tbytes(s.out);
                        int pi = E.prof_begin(st, SEG_K_STEM, tb * x.ndy, 0.0);
                        launch_stemx(x, 2, E.ndim, E.dtype, nullptr, nullptr, st);
                        E.prof_end(st, pi);
                        GnBwdFinArgs fin[2];
                        int nfin = 0;
                        for (int ui : {s.ua, s.ub}) {
                            if (ui < 0) continue;
                            const Step& u = E.steps[ui];
                            GnBwdFinArgs f{};
                            f.Q = (double*)(E.ws + u.Q); f.stats = (double*)(E.ws + u.stats);
                            f.gamma = E.p + E.params[u.gn_w].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.dgamma = E.g + E.params[u.gn_w].off; f.dbeta = E.g + E.params[u.gn_b].off;
                            f.dbias = u.b >= 0 ? E.g + E.params[u.b].off : nullptr;
                            f.coef = (float*)(E.ws + u.coef);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(0);
                            fin[nfin++] = f;
                        }
                        launch_gn_bwd_finalize(fin[0], st, nfin > 1 ? &fin[1] : nullptr);  // both branches: one launch
                        pi = E.prof_begin(st, SEG_K_STEM, tb * x.ndy, 0.0);
                        launch_stemx(x, 3, E.ndim, E.dtype, E.g + E.params[E.steps[s.ua].w].off,
                                     s.ub >= 0 ? E.g + E.params[E.steps[s.ub].w].off : nullptr, st);
                        E.prof_end(st, pi);
                    });
                    continue;
                }
                const bool dual = s.ua >= 0 && s.ub >= 0 && E.dual_gn_bwd &&
                                  !gn_bwd_group_eligible(E.tens[E.steps[s.ua].raw].C, E.vol(E.tens[E.steps[s.ua].raw].lvl), (int)E.esz());
                if (dual) {
                    // both branches of the VNet input block (one GroupNorm module applied twice, networks/VNet3d.py:36-41) receive
                    // the SAME gradient sources: one reduce and one apply pass read them once for both (14 -> 10 tensor passes)
                    Step& ua = E.steps[s.ua];
                    Step& ub = E.steps[s.ub];
                    ua.draw = new_grad(ua.raw);
                    ub.draw = new_grad(ub.raw);
                    E.bwd_writes.push_back({ua.gn_w, ua.gn_b, ua.b, ub.gn_w, ub.gn_b, ub.b});
                    E.bwd_ops.push_back([this_ = &E, uia = s.ua, uib = s.ub, gl, fill](hipStream_t st) {
                        seg_engine& E = *this_;
                        GnBwdArgs a, b;
                        GnBwdFinArgs fa{}, fb{};
                        fill(E, uia, gl, a, fa);
                        fill(E, uib, gl, b, fb);
                        a.r2 = b.r; a.scale2 = b.scale; a.shift2 = b.shift; a.Q2 = b.Q; a.coef2 = b.coef; a.dr2 = b.dr;
                        const double tb = E.tbytes(E.steps[uia].raw);
                        int pi = E.prof_begin(st, SEG_K_GN_BWD_REDUCE, tb * (a.ndy + 2), 0.0);
                        launch_gn_bwd_reduce(a, E.dtype, st);
                        E.prof_end(st, pi);
                        const bool fold = E.use_fold && a.C <= 256;
                        if (!fold) { launch_gn_bwd_finalize(fa, st); launch_gn_bwd_finalize(fb, st); }
                        pi = E.prof_begin(st, SEG_K_GN_BWD_APPLY, tb * (a.ndy + 4), 0.0);
                        launch_gn_bwd_apply(a, E.dtype, st, fold ? &fa : nullptr, fold ? &fb : nullptr);
                        E.prof_end(st, pi);
                    });
                } else
                for (int ui : {s.ua, s.ub}) {
                    if (ui < 0) continue;
                    Step& u = E.steps[ui];
                    u.draw = new_grad(u.raw);
                    E.bwd_writes.push_back({u.gn_w, u.gn_b, u.b});      // gamma/beta and (analytically) the conv bias
                    E.bwd_ops.push_back([this_ = &E, ui, gl, fill, asi = si](hipStream_t st) {
                        seg_engine& E = *this_;
                        const Step& u = E.steps[ui];
                        const Ten& r = E.tens[u.raw];
                        GnBwdArgs a;
                        GnBwdFinArgs f{};
                        fill(E, ui, gl, a, f);
                        if (E.use_coop && gn_bwd_coop_eligible(a, (int)E.esz())) {
                            // 24^3 ... 6^3 levels: reduce + finalize + apply in one launch on ~one workgroup per CU (each tensor read once)
                            const int pg = E.prof_begin(st, SEG_K_GN_GROUP, E.tbytes(u.raw) * (a.ndy + 2), 0.0);
                            launch_gn_bwd_coop(a, f, E.dtype, st);
                            E.prof_end(st, pg);
                            return;
                        }
                        if (gn_bwd_group_eligible(r.C, a.V, (int)E.esz())) {
                            const int pg = E.prof_begin(st, SEG_K_GN_GROUP, E.tbytes(u.raw) * (2 * a.ndy + 3), 0.0);
                            launch_gn_bwd_group(a, f, E.dtype, st);
                            E.prof_end(st, pg);
                            return;
                        }
                        int pi;
                        if (E.steps[asi].rq_fused) a.rep_q = f.rep_q = 0;      // the sums came with the data-gradient launch that wrote dy[0], spread over all STAT_REP replicas
                        else {
                            pi = E.prof_begin(st, SEG_K_GN_BWD_REDUCE, E.tbytes(u.raw) * (a.ndy + 1), 0.0);
                            launch_gn_bwd_reduce(a, E.dtype, st);
                            E.prof_end(st, pi);
                        }
                        const bool fold = E.use_fold && a.C <= 256;
                        if (!fold) launch_gn_bwd_finalize(f, st);
                        pi = E.prof_begin(st, SEG_K_GN_BWD_APPLY, E.tbytes(u.raw) * (a.ndy + 2), 0.0);
                        launch_gn_bwd_apply(a, E.dtype, st, fold ? &f : nullptr, nullptr);
                        E.prof_end(st, pi);
                    });
                }
            } else {   // UNIT: weight gradient + data gradient given d(raw)
                if (s.fused_stem) continue;      // weight gradients come out of the fused input block (the ACT op above)
                int draw = s.draw;
                if (s.gn_w < 0) {
                    // plain ConvTranspose (UNet up-conv): d(raw) is the (single) gradient of its output tensor
                    std::vector<int> gl = E.tens[s.raw].grads;
                    if (gl.size() != 1) { g_err = "internal: plain conv output needs exactly one gradient"; return; }
                    draw = gl[0];
                    if (E.tens[draw].virt) E.head_din_needed = true;
                }
                if (draw < 0) { g_err = "internal: unit without output gradient"; return; }
                const bool need_dg0 = !E.tens[s.in0].image;
                int g0 = -1, g1 = -1;
                if (need_dg0) { g0 = new_grad(s.in0); E.tens[s.in0].grads.push_back(g0); }
                if (s.in1 >= 0) { g1 = new_grad(s.in1); E.tens[s.in1].grads.push_back(g1); }
                E.bwd_writes.push_back({s.w, s.gn_w < 0 ? s.b : -1});
                E.bwd_ops.push_back([this_ = &E, si, draw, g0, g1](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& i0 = E.tens[s.in0];
                    const Ten& ro = E.tens[s.raw];
                    const int li = i0.lvl, lo = ro.lvl;
                    const int T = (s.ck == CK_K3 || s.ck == CK_STEM3) ? (E.ndim == 3 ? 27 : 9)
                                  : (s.ck == CK_K2S2 || s.ck == CK_KT) ? (E.ndim == 3 ? 8 : 4) : 1;
                    // ---- bias gradient of convs without GroupNorm (the UNet up-convs): a column sum of d(raw).  On the main queue: the second queue carries the
                    // critical tail of the UNet steps (moved there in round 5: UNet3d 2 x 128^3 4.46-4.49 vs 4.38-4.44 ms, profiles/r05_colsum_ab.log)
                    if (s.gn_w < 0 && s.b >= 0)
                        launch_colsum(E.ws + E.tens[draw].off, E.g + E.params[s.b].off, (long long)E.N * E.vol(lo), s.Cout, E.dtype, st);
                    if (s.ck == CK_K3) {
                        // halo-tile kernels: weight gradient (deterministic two-stage reduction) + data gradient(s)
                        const double fl = 2.0 * E.N * E.vol(lo) * (E.ndim == 3 ? 27 : 9) * s.Cin * s.Cout;
                        E.defer_wgrad(st, [this_, si, draw, fl, lo](hipStream_t ws_) {
                            seg_engine& E = *this_;
                            const Step& s = E.steps[si];
                            const Ten& i0 = E.tens[s.in0];
                            const int pi = E.prof_begin(ws_, SEG_K_WGRAD3, E.tbytes(draw) + E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0), fl);
                            Wgrad3Reduce rd;
                            float* slot = E.w3_slot(lo, ws_);
                            launch_wgrad3(E.ws + E.tens[draw].off, E.ws + i0.off, slot, E.g + E.params[s.w].off,
                                          E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, s.Cin, E.ndim, E.dtype, ws_,
                                          s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C, s.cin_par, &rd);
                            E.w3_pending.push_back(rd);
                            if (E.w3_mode == 0 || !E.use_side) E.flush_w3();
                            E.prof_end(ws_, pi);
                        }, E.tbytes(draw), lo);
                        int pi;
                        if (g0 >= 0) {
                            pi = E.prof_begin(st, conv3_class(E.dim_w(lo), s.Cout), E.tbytes(draw) + E.tbytes(g0), fl * i0.C / s.Cin);
                            if (s.x_dg0 >= 0)
                                launch_conv3x(s.x_dg0, E.ws + E.tens[draw].off, nullptr, 0, E.ws + s.wp_dg0, nullptr, E.ws + E.tens[g0].off, nullptr,
                                              E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, i0.C, E.ndim, E.dtype, st);
                            else
                            launch_conv3(E.ws + E.tens[draw].off, E.ws + s.wp_dg0, nullptr, E.ws + E.tens[g0].off, nullptr, E.N,
                                         E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, i0.C, E.ndim, E.dtype, st);
                            E.prof_end(st, pi);
                        }
                        if (g1 >= 0) {
                            const int C1 = E.tens[s.in1].C;
                            pi = E.prof_begin(st, conv3_class(E.dim_w(lo), s.Cout), E.tbytes(draw) + E.tbytes(g1), fl * C1 / s.Cin);
                            if (s.x_dg1 >= 0)
                                launch_conv3x(s.x_dg1, E.ws + E.tens[draw].off, nullptr, 0, E.ws + s.wp_dg1, nullptr, E.ws + E.tens[g1].off, nullptr,
                                              E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, C1, E.ndim, E.dtype, st);
                            else
                            launch_conv3(E.ws + E.tens[draw].off, E.ws + s.wp_dg1, nullptr, E.ws + E.tens[g1].off, nullptr, E.N,
                                         E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, C1, E.ndim, E.dtype, st);
                            E.prof_end(st, pi);
                        }
                        return;
                    }
                    if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                        // the image stems close the backward pass: nothing is left on the main stream to overlap with, so the
                        // 1^d stem (own scratch) runs on the main stream next to the 3^d stem on the side stream
                        auto run = [this_, si, draw](hipStream_t ws_) {
                            seg_engine& E = *this_;
                            const Step& s = E.steps[si];
                            const Ten& i0 = E.tens[s.in0];
                            // the stems run on the main stream (in order there) with their own scratch; the shared partial buffer belongs to
                            // whatever the weight-gradient stream is still reducing
                            const size_t scratch = E.off_partial_stem1;
                            const int pi = E.prof_begin(ws_, SEG_K_STEM, E.tbytes(draw) + E.tbytes(s.in0), 0.0);
                            launch_stem_wgrad(E.ws + E.tens[draw].off, E.ws + i0.off, (float*)(E.ws + scratch), E.g + E.params[s.w].off,
                                              E.N, E.dim_d(0), E.dim_h(0), E.dim_w(0), i0.C, s.Cout, s.ck == CK_STEM1, E.ndim, E.dtype, ws_);
                            E.prof_end(ws_, pi);
                        };
                        // step-24 trace: with the 3^d stem on the side stream the main stream idled 256 us at the end of every step
                        // behind wgrad3(16ch@96^3) + the 1^d concat wgrad + this kernel; both stems now run on the main stream
                        E.flush_side(st);
                        run(st);
                        return;
                    }
                    // ---- weight gradient
                    E.defer_wgrad(st, [this_, si, draw](hipStream_t ws_) {
                        seg_engine& E = *this_;
                        const Step& s = E.steps[si];
                        WgradArgs w = make_wgrad_args(E, s, draw);
                        const int pi = E.prof_begin(ws_, SEG_K_WGRAD_GENERIC,
                                                    E.tbytes(draw) + E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0), 0.0);
                        launch_wgrad(w, (float*)(E.ws + E.cur_partial), E.dtype, ws_, s.cin_par);
                        E.prof_end(ws_, pi);
                    }, E.tbytes(draw), lo < li ? lo : li);
                    // ---- data gradient(s)
                    if (g0 < 0 && g1 < 0) return;
                    ConvArgs a{};
                    a.in0 = E.ws + E.tens[draw].off; a.C0 = s.Cout; a.in1 = nullptr; a.C1 = 0;
                    a.bias = nullptr; a.stats = nullptr; a.N = E.N;
                    if (s.ck == CK_K2S2) {
                        // d_in[2o+a][ci] = sum_co draw[o][co] W[co][ci][a] : scatter GEMM over coarse rows
                        a.scatter = 1; a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off;
                        a.ID = a.OD = E.dim_d(lo); a.IH = a.OH = E.dim_h(lo); a.IW = a.OW = E.dim_w(lo);
                        a.FD = E.dim_d(li); a.FH = E.dim_h(li); a.FW = E.dim_w(li);
                        a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
                        a.taps = make_taps(E.ndim, 2, 0);
                        a.Cout = s.Cin; a.K = s.Cout; a.Ngemm = a.taps.n * s.Cin; a.Kpad = (a.K + 31) / 32 * 32;
                        launch_conv_igemm(a, E.dtype, st, STAT_REP);
                    } else if (s.ck == CK_KT) {
                        // d_X[i][ci] = sum_{a,co} dY[2i+a][co] Wt[ci][co][a] : gather, stride 2 over the fine gradient
                        a.scatter = 0; a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off;
                        a.ID = E.dim_d(lo); a.IH = E.dim_h(lo); a.IW = E.dim_w(lo);
                        a.OD = E.dim_d(li); a.OH = E.dim_h(li); a.OW = E.dim_w(li);
                        a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
                        a.taps = make_taps(E.ndim, 2, 0);
                        a.Cout = s.Cin; a.Ngemm = s.Cin; a.K = a.taps.n * s.Cout; a.Kpad = (a.K + 31) / 32 * 32;
                        launch_conv_igemm(a, E.dtype, st, STAT_REP);
                    } else {
                        // conv 3^d / 1^d: gather conv of d(raw) with flipped taps, once per concat source
                        a.scatter = 0;
                        a.ID = a.OD = E.dim_d(lo); a.IH = a.OH = E.dim_h(lo); a.IW = a.OW = E.dim_w(lo);
                        a.sd = a.sh = a.sw = 1;
                        const int k = s.ck == CK_K3 ? 3 : 1;
                        a.taps = make_taps(E.ndim, k, k == 3 ? 1 : 0);
                        a.K = a.taps.n * s.Cout; a.Kpad = (a.K + 31) / 32 * 32;
                        if (g0 >= 0 && g1 >= 0 && s.dual_dg) {
                            // 1^d conv on a concat: both data-gradients from ONE pass over d(raw) (113 MB at the 96^3 level)
                            ConvArgs b;
                            if (make_dual_dgrad_args(E, s, draw, g0, g1, b)) {
                                if (s.vact_unit >= 0) {
                                    const Step& pu = E.steps[s.vact_unit];
                                    bool rq = false;
                                    for (const Step& A : E.steps) if (A.type == ST_ACT && A.vact && A.ua == s.vact_unit) rq = A.rq_fused;
                                    if (rq) {        // ... and the GroupNorm-backward sums of the up-conv unit whose (virtual) activation is the first source
                                        b.rq_r = E.ws + E.tens[pu.raw].off; b.rq_scale = (const float*)(E.ws + pu.scale); b.rq_shift = (const float*)(E.ws + pu.shift);
                                        b.rq_Q = (double*)(E.ws + pu.Q);
                                    }
                                }
                                launch_conv_igemm(b, E.dtype, st, STAT_REP);
                                return;
                            }
                        }
                        if (g0 >= 0) {
                            a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off; a.Cout = a.Ngemm = E.tens[s.in0].C;
                            launch_conv_igemm(a, E.dtype, st, STAT_REP);
                        }
                        if (g1 >= 0) {
                            a.w = E.ws + s.wp_dg1; a.out = E.ws + E.tens[g1].off; a.Cout = a.Ngemm = E.tens[s.in1].C;
                            launch_conv_igemm(a, E.dtype, st, STAT_REP);
                        }
                    }
                });
            }
        }
        E.ws_bytes = align_up(cur, 4096);
        E.planned = true;
        (void)dt;
    }
};

}  // namespace

namespace segi {
void build_network(seg_engine& e, int net_kind) {
    Builder b(e);
    if (net_kind == SEG_NET_VNET) b.build_vnet(); else b.build_unet();
}
void plan_engine(seg_engine& e) {
    Planner pl(e);
    pl.plan();
}
}  // namespace segi
