// Host side of libdffw.so: the layer table (weight contract), BatchNorm folding + MFMA-fragment
// weight packing, the static workspace arena and the whole-graph executor of DFF_net.forward
// (reference Depth_Estimation_Test/Depth_Estimation_Network.py:74-127) and of the End_to_End variant
// (End_to_End/End_to_End.py: alignment network + FOV warp in front of the same DFF_net), behind the C ABI
// of include/dffw.h.  All arithmetic of the forward runs in the gfx950 kernels of dffw_kernels.hip;
// nothing here touches activation values.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dffw.h"
#include "dffw_conv_roll.h"
#include "dffw_srd_roll.h"
#include "dffw_stem.h"
#include "dffw_conv_tile.h"
#include "dffw_internal.h"

namespace dffw {

// ---- errors ------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local std::string g_last_kernel;   // dffw_last_conv_kernel()
static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace dffw
int dffw_fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    dffw::g_err = buf;
    return code;
}
namespace dffw {
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return fail(DFFW_EHIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- layer table -------------------------------------------------------------------------------
struct LayerDef {
    std::string conv, bn;  // state-dict prefixes ("" = no BatchNorm)
    int cin, cout;
    int kd, kh, kw;
    int sh, sw;            // stride over rows/cols (slice stride is always 1 in this network)
    int pd, ph, pw;
    int dh, dw;            // dilation over rows/cols
    bool transposed;       // ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1)
    bool live;
    bool bias;
    std::string shortcut = "";  // prefix of a bias-free 1x1x1 stride-1 conv over a SECOND input whose result is added to this
                                // layer's: folded into this layer's weights as centre-tap columns of a channel concat
    bool folded = false;        // this layer is such a shortcut: it is never launched on its own
    int head_split = 0;         // > 0: first conv of an alignment head over [ref (C) | cur (C) | flow (2)], C = head_split:
                                // also packed as "<name>#ref" (ref channels, BatchNorm scale only) and "<name>#cur" (the rest)
};

struct ParamInfo {
    std::string name;
    int64_t shape[5];
    int ndim;
    int flags;  // bit0 buffer, bit1 int64 counter, bit2 dead
};

class Table {
  public:
    std::vector<LayerDef> layers;
    std::vector<ParamInfo> params;
    std::map<std::string, int> by_name;

    void conv(const std::string &key, int cin, int cout, int kd, int kh, int kw, int s, int pd, int ph, int pw, int dil,
              bool bn, bool live = true) {
        // `key` is the prefix of a reference convbn_3d pair when bn (conv = key.0, BatchNorm = key.1,
        // DEN.py:286-289), else the conv's own prefix
        LayerDef L{bn ? key + ".0" : key, bn ? key + ".1" : "", cin, cout, kd, kh, kw, s, s, pd, ph, pw, dil, dil, false, live, false};
        add(L);
    }
    void deconv(const std::string &key, int cin, int cout) {
        LayerDef L{key + ".0", key + ".1", cin, cout, 3, 3, 3, 2, 2, 1, 1, 1, 1, 1, true, true, false};
        add(L);
    }
    void c3(const std::string &key, int cin, int cout, int s = 1, bool bn = true) { conv(key, cin, cout, 3, 3, 3, s, 1, 1, 1, 1, bn); }

    void srd(const std::string &p, int c) {  // DEN.py:295-330
        conv(p + ".Focus_Measure.conv.0", c, c, 1, 3, 3, 1, 0, 1, 1, 1, true);
        conv(p + ".Focus_Measure.conv.2", c, c, 1, 3, 3, 1, 0, 1, 1, 1, true);
        conv(p + ".N_ch_attention.0", c, c, 3, 1, 1, 1, 1, 0, 0, 1, false);
        conv(p + ".N_ch_attention.2", c, c, 1, 1, 1, 1, 0, 0, 0, 1, false);
    }
    void efd(const std::string &p, int cin, int cout) {  // DEN.py:306-315
        c3(p + ".stride_conv", cin, cout, 2);
        c3(p + ".max_pooling.1", cin, cout, 1);
    }
    void hourglass(const std::string &p, int c) {  // DEN.py:240-264
        c3(p + ".conv0.0", 2 * c, c);
        c3(p + ".conv1.0", c, 2 * c, 2);
        conv(p + ".pre_conv.0", 2 * c, 2 * c, 1, 1, 1, 1, 0, 0, 0, 1, true, /*live=*/false);
        c3(p + ".conv2", 2 * c, 2 * c);
        c3(p + ".conv3.0", 2 * c, 2 * c, 2);
        c3(p + ".conv4.0", 2 * c, 2 * c);
        deconv(p + ".conv5", 2 * c, 2 * c);
        deconv(p + ".conv6", 2 * c, c);
    }

    void biased(const std::string &key, int cin, int cout, int kd, int kh, int kw, int pd, int ph, int pw) {
        LayerDef L{key, "", cin, cout, kd, kh, kw, 1, 1, pd, ph, pw, 1, 1, false, true, true};
        add(L);
    }
    void of_block(const std::string &p, int cin, int cout, int s) {  // resnet_block_2d_OF, End_to_End.py:135-145
        conv(p + ".conv.0", cin, cout, 1, 3, 3, s, 0, 1, 1, 1, true);
        conv(p + ".conv.2", cout, cout, 1, 3, 3, 1, 0, 1, 1, 1, true);
        conv(p + ".feature", cin, cout, 1, 1, 1, s, 0, 0, 0, 1, false);
        if (s == 1) {   // same resolution on both branches: the shortcut rides in conv.2's contraction
            layers[by_name[p + ".conv.2.0"]].shortcut = p + ".feature";
            layers[by_name[p + ".feature"]].folded = true;
        }
    }
    void alpha_head(const std::string &p, int cin, int c) {  // conv1/conv2/conv3 of FlowNetwork, End_to_End.py:37-69
        conv(p + ".0", cin, c, 1, 3, 3, 1, 0, 1, 1, 1, true);
        layers[by_name[p + ".0.0"]].head_split = (cin - 2) / 2;
        conv(p + ".2", c, c, 1, 3, 3, 1, 0, 1, 1, 1, true);
        conv(p + ".4", c, c, 1, 3, 3, 1, 0, 1, 1, 1, true);
        biased(p + ".6", c, 3, 1, 3, 3, 0, 1, 1);
    }

    // End_to_End.Network (End_to_End.py:9-12): DFF_net registered first, then optical_flow_aggregation = FlowNetwork(8)
    static Table e2e_net() {
        Table t = depth_net();
        const std::string P = "optical_flow_aggregation";
        const int C = 8;
        t.of_block(P + ".OF_feature.0", 3, C, 1);
        t.of_block(P + ".OF_feature.1", C, C, 1);
        t.of_block(P + ".OF_feature1.0", C, 2 * C, 2);
        t.of_block(P + ".OF_feature1.1", 2 * C, 2 * C, 1);
        t.of_block(P + ".OF_feature2.0", 2 * C, 4 * C, 2);
        t.of_block(P + ".OF_feature2.1", 4 * C, 4 * C, 1);
        t.alpha_head(P + ".conv1", 8 * C + 2, 8 * C);
        t.alpha_head(P + ".conv2", 4 * C + 2, 4 * C);
        t.alpha_head(P + ".conv3", 2 * C + 2, 2 * C);
        return t;
    }

    static Table depth_net() {
        Table t;
        const std::string P = "DFF_net";
        t.conv(P + ".FM_measure.Focus_extraction.0", 3, 8, 1, 9, 9, 1, 0, 8, 8, 2, true);  // DEN.py:135
        t.srd(P + ".FM_measure.Focus_extraction.2", 8);
        t.efd(P + ".FM_conv1.0", 8, 16);
        t.srd(P + ".FM_conv1.1", 16);
        t.efd(P + ".FM_conv2.0", 16, 32);
        t.srd(P + ".FM_conv2.1", 32);
        const std::string S = P + ".SPP_module";  // DEN.py:145-210
        const char *scales[3] = {"8", "16", "32"};
        const int widths[3] = {32, 64, 64};
        for (int i = 0; i < 3; ++i) {
            const std::string d = S + ".dres" + scales[i];
            t.c3(d + "_0.0", 32, widths[i]);
            t.c3(d + "_0.2", widths[i], widths[i]);
            t.c3(d + "_1.0", widths[i], widths[i]);
            t.c3(d + "_1.2", widths[i], widths[i]);
        }
        t.c3(S + ".conv1", 32, 64, 2, false);
        t.c3(S + ".conv2.0", 64, 64);
        t.c3(S + ".conv3", 64, 128, 2, false);
        t.c3(S + ".conv4.0", 128, 128);
        t.deconv(S + ".conv8", 128, 64);
        t.deconv(S + ".conv9", 64, 32);
        t.c3(S + ".combine1.0", 128, 64);
        t.c3(S + ".combine2.0", 192, 128);
        t.conv(S + ".redir1", 32, 32, 1, 1, 1, 1, 0, 0, 0, 1, true);
        t.conv(S + ".redir2", 64, 64, 1, 1, 1, 1, 0, 0, 0, 1, true);
        t.conv(S + ".redir3", 128, 128, 1, 1, 1, 1, 0, 0, 0, 1, true, /*live=*/false);
        t.c3(P + ".confidence.0", 32, 32);
        t.c3(P + ".confidence.2", 32, 1, 1, false);
        t.c3(P + ".dres0.0", 32, 64);
        t.c3(P + ".dres0.2", 64, 64);
        t.deconv(P + ".deconv_1", 64, 32);
        t.hourglass(P + ".dres2", 32);
        t.deconv(P + ".deconv_2", 32, 16);
        t.hourglass(P + ".dres3", 16);
        t.deconv(P + ".deconv_3", 16, 8);
        t.hourglass(P + ".dres4", 8);
        t.conv(P + ".classif1.0", 32, 1, 1, 1, 1, 1, 0, 0, 0, 1, false);
        t.conv(P + ".classif2.0", 16, 1, 1, 1, 1, 1, 0, 0, 0, 1, false);
        t.conv(P + ".classif3.0", 8, 1, 1, 1, 1, 1, 0, 0, 0, 1, false);
        return t;
    }

  private:
    void add(const LayerDef &L) {
        by_name[L.conv] = (int)layers.size();
        layers.push_back(L);
        const int dead = L.live ? 0 : 4;
        ParamInfo w{L.conv + ".weight", {0, 0, 0, 0, 0}, 5, dead};
        w.shape[0] = L.transposed ? L.cin : L.cout;
        w.shape[1] = L.transposed ? L.cout : L.cin;
        w.shape[2] = L.kd;
        w.shape[3] = L.kh;
        w.shape[4] = L.kw;
        params.push_back(w);
        if (L.bias) params.push_back(ParamInfo{L.conv + ".bias", {L.cout, 0, 0, 0, 0}, 1, dead});
        if (!L.bn.empty()) {
            params.push_back(ParamInfo{L.bn + ".weight", {L.cout, 0, 0, 0, 0}, 1, dead});
            params.push_back(ParamInfo{L.bn + ".bias", {L.cout, 0, 0, 0, 0}, 1, dead});
            params.push_back(ParamInfo{L.bn + ".running_mean", {L.cout, 0, 0, 0, 0}, 1, dead | 1});
            params.push_back(ParamInfo{L.bn + ".running_var", {L.cout, 0, 0, 0, 0}, 1, dead | 1});
            params.push_back(ParamInfo{L.bn + ".num_batches_tracked", {0, 0, 0, 0, 0}, 0, dead | 3});
        }
    }
};

static bool known_net(int net) { return net == DFFW_NET_DEPTH || net == DFFW_NET_E2E; }
static const Table &table_for(int net) {
    static const Table depth = Table::depth_net();
    static const Table e2e = Table::e2e_net();
    return net == DFFW_NET_E2E ? e2e : depth;
}

// ---- host number formats -----------------------------------------------------------------------
static uint16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float host_bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t host_f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float host_h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
static void host_split(int prec, float v, uint16_t &hi, uint16_t &lo) {
    if (prec == P_BF16X3) {
        hi = host_f2bf(v);
        lo = host_f2bf(v - host_bf2f(hi));
    } else if (prec == P_FP16) {
        hi = host_f2h(v);
        lo = 0;
        (void)host_h2f;
    } else {
        hi = host_f2bf(v);
        lo = 0;
    }
}

// ---- packed conv layer -------------------------------------------------------------------------
struct Tap {
    int dz, dy, dx;  // input offset
    int kz, ky, kx;  // which filter element
};

struct Variant {      // one launch: a regular conv, or one sub-pixel phase of a transposed conv
    int KC = 0;
    int ntaps = 0;
    int ooy = 0, oox = 0;
    TapEntry *tab = nullptr;  // device
    uint16_t *wpk = nullptr;  // device
};

struct TilePack {            // weights/taps in conv_tile's order (per pass: [stage][KC][NT][part][64][8])
    const TileCfg *cfg = nullptr;
    int nstage = 0;
    int npass = 0;
    int KC[4] = {0, 0, 0, 0};
    int ntaps[4] = {0, 0, 0, 0};
    int ooy[4] = {0, 0, 0, 0}, oox[4] = {0, 0, 0, 0};
    int *tab[4] = {nullptr, nullptr, nullptr, nullptr};
    uint16_t *wpk[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct PackedConv {
    LayerDef def;
    int nt = 1;
    float *bias = nullptr;  // device, nt*16 floats
    std::vector<Variant> variants;
    TilePack tile;
    TilePack tile_narrow;   // 3x3x3 stride-1 / transposed layers once more on the 5 x 8 x 8 block (grids at most 8 wide, Run::conv decides per call)
    TilePack tile_pair;     // the stem once more, for the pixel-pair kernel (G2P); bias_pair = its BatchNorm shift for both pixels' rows
    float *bias_pair = nullptr;
    float *w32 = nullptr;  // device fp32 [kz][cin][cout] (BatchNorm folded) for kh = kw = 1 layers: fused VALU kernels
    float *whead = nullptr;   // device fp32: the last conv of an alpha head (biased 1x3x3, 3 outputs) as [3][cin][9] weights then [3] bias,
                              // for head_tail_finish_kernel (conv + plane mean collapsed into plane sums)
    uint16_t *wroll = nullptr;  // device: the filter in conv_roll's fragment order (3x3x3 stride 1, 16 input channels, <= 16 outputs)
    bool roll_pair = false;     // ... packed for its pixel-pair variant (<= 8 output channels)
    uint16_t *wroll_k2 = nullptr;  // device: a 3x3x3 stride-1 32 -> 16 filter in conv_rollx_k2's order: [input half][conv_roll's 15 chunks]
    uint16_t *wslice32 = nullptr;  // device: a 1x3x3 32 -> 32 filter in conv_slice32's order: [9 taps][output tile][part]
    bool slice_cat = false;        // wslice64 holds a 32 -> 32 filter + folded 1x1x1 shortcut over a second 32-channel tensor in conv_slice32_cat's order
    uint16_t *wslice64 = nullptr;  // device: a 1x3x3 64 -> 64 filter in conv_slice64's order: [output tile][chunk = tap * 2 + channel half][part]; or (a 34 -> 64 `#cur` layer of
                                   // an alignment head) in its HEAD order: [output tile][9 feature chunks + 3 chunks over the flow octet][part]
    uint16_t *wrollk = nullptr;    // device: a 3x3x3 stride-1 32 / 64 -> 32 / 64 filter in conv_rollk's order: [32-channel output pair][wave][7 chunks][output tile]
    uint16_t *wrollt = nullptr;    // device: a transposed 3x3x3 32 / 64 -> 32 / 64 filter in conv_rollt's order: [32-channel output half][wave][rollt::MAXU units][part]
    uint16_t *wroll_t = nullptr;   // device: the filter in conv_roll_t's order (transposed 3x3x3, 16 -> 8 channels)
    uint16_t *wroll8 = nullptr;    // device: a 3x3x3 8 -> 16 filter (stride 1 or (1,2,2)) in conv_roll_efd's order
    uint16_t *wroll_s2 = nullptr;  // device: a 3x3x3 stride-(1,2,2) 16 -> 16 / 32 filter in conv_roll_s2's order (15 chunks per 16-channel output tile)
    uint16_t *wroll15 = nullptr;   // device: a 3x3x3 stride-1 16 -> 32 filter in the same order (the pooled branch of the fused 16-channel EFD block, conv_efd16)
    uint16_t *wroll_t32 = nullptr; // device: a transposed 3x3x3 32 -> 16 filter in conv_roll_t32's order (row phase 0: 9 chunks, then phase 1: 18)
    uint16_t *wsrd = nullptr;      // device: a 1x3x3 8 -> 8 filter in srd_roll's order (3 chunks of 4 taps x 8 channels)
    uint16_t *watt = nullptr;      // device: an 8 -> 8 attention conv (3x1x1 or 1x1x1) as srd_roll's stage-C fragments
    int cin_all = 0;       // input channels the packed layer contracts over: own (padded to 8) + folded shortcut's (padded to 8)
};

static void free_packed(PackedConv &pc) {
    if (pc.bias) (void)hipFree(pc.bias);
    for (auto &v : pc.variants) {
        if (v.tab) (void)hipFree(v.tab);
        if (v.wpk) (void)hipFree(v.wpk);
    }
    pc.variants.clear();
    pc.bias = nullptr;
    for (TilePack *tp : {&pc.tile, &pc.tile_pair, &pc.tile_narrow}) {
        for (int i = 0; i < 4; ++i) {
            if (tp->tab[i]) (void)hipFree(tp->tab[i]);
            if (tp->wpk[i]) (void)hipFree(tp->wpk[i]);
            tp->tab[i] = nullptr;
            tp->wpk[i] = nullptr;
        }
        tp->cfg = nullptr;
    }
    if (pc.bias_pair) (void)hipFree(pc.bias_pair);
    pc.bias_pair = nullptr;
    if (pc.w32) (void)hipFree(pc.w32);
    pc.w32 = nullptr;
    if (pc.whead) (void)hipFree(pc.whead);
    pc.whead = nullptr;
    if (pc.wroll) (void)hipFree(pc.wroll);
    pc.wroll = nullptr;
    if (pc.wroll_t) (void)hipFree(pc.wroll_t);
    pc.wroll_t = nullptr;
    if (pc.wroll_k2) (void)hipFree(pc.wroll_k2);
    pc.wroll_k2 = nullptr;
    if (pc.wrollk) (void)hipFree(pc.wrollk);
    pc.wrollk = nullptr;
    if (pc.wrollt) (void)hipFree(pc.wrollt);
    pc.wrollt = nullptr;
    if (pc.wslice32) (void)hipFree(pc.wslice32);
    pc.wslice32 = nullptr;
    if (pc.wslice64) (void)hipFree(pc.wslice64);
    pc.wslice64 = nullptr;
    if (pc.wroll8) (void)hipFree(pc.wroll8);
    pc.wroll8 = nullptr;
    if (pc.wroll_t32) (void)hipFree(pc.wroll_t32);
    pc.wroll_t32 = nullptr;
    if (pc.wroll_s2) (void)hipFree(pc.wroll_s2);
    pc.wroll_s2 = nullptr;
    if (pc.wroll15) (void)hipFree(pc.wroll15);
    pc.wroll15 = nullptr;
    if (pc.wsrd) (void)hipFree(pc.wsrd);
    pc.wsrd = nullptr;
    if (pc.watt) (void)hipFree(pc.watt);
    pc.watt = nullptr;
}

// weight: PyTorch layout.  bn: gamma|beta|mean|var (4*cout) or null.  conv_bias: cout or null.
// shortcut_w: (cout, shortcut_cin) weights of a folded 1x1x1 shortcut over a second input, or null.
static int pack_conv(const LayerDef &L, int prec, const float *weight, const float *bn, const float *conv_bias,
                     PackedConv &pc, const float *shortcut_w = nullptr, int shortcut_cin = 0) {
    pc.def = L;
    pc.nt = conv_nt_for(L.cout);
    const int parts = prec_parts(prec);
    const int cin_own = (L.cin + 7) / 8 * 8;
    const int cin_pad = cin_own + (shortcut_w ? (shortcut_cin + 7) / 8 * 8 : 0);   // channels of the (virtual) input concat
    const int c8n = cin_pad / 8;
    pc.cin_all = cin_pad;

    // fold BatchNorm (eval mode, eps 1e-5): y = conv(x)*scale + shift
    std::vector<double> scale(L.cout, 1.0), shift(L.cout, 0.0);
    for (int c = 0; c < L.cout; ++c) {
        if (bn) {
            const double g = bn[c], b = bn[L.cout + c], m = bn[2 * L.cout + c], v = bn[3 * L.cout + c];
            scale[c] = g / std::sqrt(v + 1e-5);
            shift[c] = b - m * scale[c];
        }
        if (conv_bias) shift[c] += conv_bias[c] * scale[c];
    }
    std::vector<float> bias(pc.nt * 16, 0.f);
    for (int c = 0; c < L.cout; ++c) bias[c] = (float)shift[c];
    HIPCHK(hipMalloc((void **)&pc.bias, bias.size() * sizeof(float)));
    HIPCHK(hipMemcpy(pc.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    if (L.cout == 8 && pc.nt == 1) {   // pixel-pair kernels: rows 8-15 are the second pixel's 8 channels
        std::vector<float> b2(16);
        for (int c = 0; c < 16; ++c) b2[c] = (float)shift[c & 7];
        HIPCHK(hipMalloc((void **)&pc.bias_pair, b2.size() * sizeof(float)));
        HIPCHK(hipMemcpy(pc.bias_pair, b2.data(), b2.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // The stem reads the paired-pixel (W+2)-wide volume written by stack_in: pixel p lives in the first half
    // of record p+2, so every x-offset of its taps is shifted by +2.
    const bool stem = (L.kd == 1 && L.kh == 9 && L.kw == 9 && L.dh == 2 && L.pd == 0 && L.ph == 8 && L.sh == 1 && L.cin == 3);
    // tap lists
    std::vector<std::vector<Tap>> tapsets;
    std::vector<std::pair<int, int>> phase;
    if (!L.transposed) {
        std::vector<Tap> taps;
        for (int kz = 0; kz < L.kd; ++kz)
            for (int ky = 0; ky < L.kh; ++ky)
                for (int kx = 0; kx < L.kw; ++kx)
                    taps.push_back(Tap{kz - L.pd, ky * L.dh - L.ph, kx * L.dw - L.pw + (stem ? 2 : 0), kz, ky, kx});
        tapsets.push_back(taps);
        phase.push_back({0, 0});
    } else {
        // out[oz,oy,ox] = sum in[iz,iy,ix] * w[kz,ky,kx] with oz = iz-1+kz, oy = 2*iy-1+ky, ox = 2*ix-1+kx.
        // For output parity p along a stride-2 axis (o = 2*g + p): p=0 uses k=1 at i=g; p=1 uses k=0 at
        // i=g+1 and k=2 at i=g.  Never materialise the zero-inserted input.
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                std::vector<Tap> taps;
                for (int kz = 0; kz < 3; ++kz)
                    for (int ky = 0; ky < 3; ++ky) {
                        if ((ky & 1) == py) continue;  // py=0 -> ky odd only; py=1 -> ky even only
                        for (int kx = 0; kx < 3; ++kx) {
                            if ((kx & 1) == px) continue;
                            const int dy = (ky == 0) ? 1 : 0, dx = (kx == 0) ? 1 : 0;
                            taps.push_back(Tap{1 - kz, dy, dx, kz, ky, kx});
                        }
                    }
                tapsets.push_back(taps);
                phase.push_back({py, px});
            }
    }

    const int kvol = L.kd * L.kh * L.kw;
    auto wval = [&](int cout, int cin, const Tap &t) -> double {
        if (cin >= cin_own) {   // folded shortcut: its own weight on the centre tap (not scaled by this layer's BatchNorm)
            const int ce = cin - cin_own;
            return (ce < shortcut_cin && t.dz == 0 && t.dy == 0 && t.dx == 0) ? (double)shortcut_w[(int64_t)cout * shortcut_cin + ce] : 0.0;
        }
        if (cin >= L.cin) return 0.0;
        const int64_t kidx = ((int64_t)t.kz * L.kh + t.ky) * L.kw + t.kx;
        const int64_t i = L.transposed ? ((int64_t)cin * L.cout + cout) * kvol + kidx : ((int64_t)cout * L.cin + cin) * kvol + kidx;
        return (double)weight[i] * scale[cout];
    };

    for (size_t vi = 0; vi < tapsets.size(); ++vi) {
        const auto &taps = tapsets[vi];
        Variant v;
        v.ooy = phase[vi].first;
        v.oox = phase[vi].second;
        v.ntaps = (int)taps.size();
        const int K8 = (int)taps.size() * c8n;
        v.KC = (K8 + 3) / 4;
        std::vector<TapEntry> tab(v.KC * 4);
        for (int k8 = 0; k8 < v.KC * 4; ++k8) {
            if (k8 < K8) {
                const Tap &t = taps[k8 / c8n];
                tab[k8] = TapEntry{t.dz, t.dy, t.dx, (k8 % c8n) * 8};
            } else {
                tab[k8] = TapEntry{0, 0, 0, -1};
            }
        }
        std::vector<uint16_t> wpk((size_t)v.KC * pc.nt * parts * 64 * 8, 0);
        for (int kc = 0; kc < v.KC; ++kc)
            for (int nt = 0; nt < pc.nt; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int cout = nt * 16 + (lane & 15);
                        const int k = kc * 32 + (lane >> 4) * 8 + j;
                        const int tapi = k / cin_pad, cin = k % cin_pad;
                        float val = 0.f;
                        if (cout < L.cout && tapi < (int)taps.size()) val = (float)wval(cout, cin, taps[tapi]);
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)kc * pc.nt + nt) * parts) * 512 + (size_t)lane * 8 + j;
                        wpk[base] = hi;
                        if (parts == 2) wpk[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&v.tab, tab.size() * sizeof(TapEntry)));
        HIPCHK(hipMemcpy(v.tab, tab.data(), tab.size() * sizeof(TapEntry), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void **)&v.wpk, wpk.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(v.wpk, wpk.data(), wpk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        pc.variants.push_back(v);
    }

    if (!L.transposed && L.kh == 1 && L.kw == 1 && L.cout <= 32 && L.cin <= 32) {
        std::vector<float> w32((size_t)L.kd * L.cin * L.cout);
        for (int kz = 0; kz < L.kd; ++kz)
            for (int ci = 0; ci < L.cin; ++ci)
                for (int co = 0; co < L.cout; ++co) w32[((size_t)kz * L.cin + ci) * L.cout + co] = (float)wval(co, ci, Tap{0, 0, 0, kz, 0, 0});
        HIPCHK(hipMalloc((void **)&pc.w32, w32.size() * sizeof(float)));
        HIPCHK(hipMemcpy(pc.w32, w32.data(), w32.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    if (!L.transposed && L.bias && !bn && conv_bias && L.cout == 3 && L.kd == 1 && L.kh == 3 && L.kw == 3 && L.sh == 1 && L.ph == 1 && L.pw == 1 &&
        L.dh == 1 && L.cin % 8 == 0 && L.cin <= 64 && !shortcut_w) {
        std::vector<float> wh((size_t)3 * L.cin * 9 + 3);
        memcpy(wh.data(), weight, (size_t)3 * L.cin * 9 * sizeof(float));      // PyTorch (3, cin, 1, 3, 3) is already [c][ci][dy][dx]
        memcpy(wh.data() + (size_t)3 * L.cin * 9, conv_bias, 3 * sizeof(float));
        HIPCHK(hipMalloc((void **)&pc.whead, wh.size() * sizeof(float)));
        HIPCHK(hipMemcpy(pc.whead, wh.data(), wh.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // ---- second packing for the LDS-tiled kernel, when a configuration covers this geometry ----------
    int geo = -1;
    if (L.transposed) geo = G3T;
    else if (L.kd == 3 && L.kh == 3 && L.kw == 3 && L.dh == 1 && L.pd == 1 && L.ph == 1 && L.sh == 1) geo = G3S1;
    else if (L.kd == 3 && L.kh == 3 && L.kw == 3 && L.dh == 1 && L.pd == 1 && L.ph == 1 && L.sh == 2) geo = G3S2;
    else if (L.kd == 1 && L.kh == 3 && L.kw == 3 && L.dh == 1 && L.pd == 0 && L.ph == 1 && L.sh == 1) geo = G2S1;
    else if (L.kd == 1 && L.kh == 3 && L.kw == 3 && L.dh == 1 && L.pd == 0 && L.ph == 1 && L.sh == 2) geo = G2S2;
    int cin_t = cin_pad;  // channels the tiled kernel contracts over (padding channels carry zero weights)
    if (stem) {
        // paired-pixel input (stack_in): record q = RGB(q-2) | RGB(q).  Taps (ky, jx) for jx in {0,2,4,6,8}
        // read record x+2*jx-6 and carry the weights of x-taps jx (channels 0..2) and jx+1 (channels
        // 4..6; zero for the non-existent tap 9).
        geo = G2D;
        cin_t = 8;
        tapsets.assign(1, std::vector<Tap>());
        for (int ky = 0; ky < 9; ++ky)
            for (int jx = 0; jx < 9; jx += 2) tapsets[0].push_back(Tap{0, 2 * ky - 8, 2 * jx - 6, 0, ky, jx});
    }
    auto wval_t = [&](int cout, int cin, const Tap &t) -> double {
        if (!stem) return wval(cout, cin, t);
        if ((cin & 3) == 3) return 0.0;
        Tap u = t;
        if (cin >= 4) {
            if (t.kx + 1 > 8) return 0.0;
            u.kx = t.kx + 1;
        }
        return wval(cout, cin & 3, u);
    };
    if (geo >= 0 && cin_t % 8 == 0) {
        int cg = (geo == G3S2 || geo == G2S2) ? 8 : (cin_t % 16 == 0 ? 16 : 8);
        // transposed conv: its 4 sub-pixel passes share one LDS image only when the whole contraction depth is
        // staged at once, so 32-channel groups (one fill instead of 4 x 2) where an instantiation exists
        if (geo == G3T && cin_t % 32 == 0 && tile_cfg_find(geo, pc.nt, 32)) cg = 32;
        // stride-(1,2,2) 3x3x3 over 64 channels (dres2.conv3): 16-channel stages on a 4 x 4 x 8 tile instead of eight 8-channel
        // stages on 5 x 4 x 16 (every stage re-fetches the 128-byte lines it takes a piece of): -18 %.  Measured on the 16- and
        // 32-channel stride-2 layers too: +25 % / +9 % SLOWER (smaller tile, more halo, 4-slice tiles on 10 slices) -- not used there.
        if (geo == G3S2 && cin_t % 64 == 0 && tile_cfg_find(geo, pc.nt, 16)) cg = 16;
        // per-slice 1x3x3 over 32 channels: ONE 32-channel stage per tile (each 128-byte pixel line is fetched once instead of
        // half of it per 16-channel stage -- the memory side moves whole 128-byte lines, profiles/r02_fetch_size_calibration.txt)
        if (geo == G2S1 && cin_t % 32 == 0 && tile_cfg_find(geo, pc.nt, 32)) cg = 32;   // +4..14 % on those layers
        // wide (8-wave, 640-point) tile wherever an instantiation exists (dffw_conv_tile.hip lists what was measured)
        // (the pack-time switches DFFW_NO_WIDE / DFFW_NO_CG32 / DFFW_NO_S2_CG16 / DFFW_STEM_NARROW / DFFW_NO_ROLL_PAIR / DFFW_NO_ROLL_T were retired in round 5: their
        // alternatives lost every A/B of rounds 1-4, profiles/r04_ab_forward_switches.txt and the rounds before)
        const bool wide = true;
        // pair: the stem's pixel-pair form (G2P) -- result rows 8-15 carry the filter as pixel x+2 sees the same records, and the
        // LDS image keeps only the footprint columns = 0,1 mod 4 (tap offsets in packed columns)
        auto pack_tile = [&](TilePack &tp, const TileCfg *cfg, int cg, bool pair) -> int {
            const GeoInfo gi = geo_info(cfg->geo);
            tp.cfg = cfg;
            tp.nstage = cin_t / cg;
            tp.npass = (int)tapsets.size();
            const int cg8 = cg / 8;
            for (int ps = 0; ps < tp.npass; ++ps) {
                // pair form (round 6): chunks 0-8 = filter row ky with the pair columns jx = 0, 2, 4, 6 as its four K octets, chunks 9-11 = the last pair column
                // (jx = 8) of rows 0-3, 4-7, 8.  Output rows y and y + 2 then read the SAME operand fragment for (y + 2, ky) and (y, ky + 1) -- the dilation is 2 --,
                // which stem_pipe loads once (its LDS port is the kernel's busiest unit: 0.65); conv_tile's pair-form kernel just follows the table
                std::vector<Tap> ptaps;
                if (pair && tapsets[ps].size() == 45) {
                    for (int ky = 0; ky < 9; ++ky)
                        for (int ji = 0; ji < 4; ++ji) ptaps.push_back(tapsets[ps][ky * 5 + ji]);
                    for (int ky = 0; ky < 9; ++ky) ptaps.push_back(tapsets[ps][ky * 5 + 4]);
                }
                const auto &taps = ptaps.empty() ? tapsets[ps] : ptaps;
                const int K8 = (int)taps.size() * cg8;
                const int KC = (K8 + 3) / 4;
                tp.KC[ps] = KC;
                tp.ntaps[ps] = (int)taps.size();
                tp.ooy[ps] = phase[ps].first;
                tp.oox[ps] = phase[ps].second;
                std::vector<int> tab(KC * 4, 0);
                for (int k8 = 0; k8 < K8; ++k8) {
                    const Tap &tpp = taps[k8 / cg8];
                    const int dzz = tpp.dz - gi.minz, dyy = tpp.dy - gi.miny, dxx = tpp.dx - gi.minx;
                    const int lx = pair ? dxx / 2 : ((gi.s == 2) ? ((dxx & 1) * (cfg->fxl / 2) + (dxx >> 1)) : dxx);
                    tab[k8] = ((dzz * cfg->fy + dyy) * cfg->fxl + lx) * (cg * 2) + (k8 % cg8) * 16;
                }
                std::vector<uint16_t> wpk((size_t)tp.nstage * KC * pc.nt * parts * 512, 0);
                for (int st = 0; st < tp.nstage; ++st)
                    for (int kc = 0; kc < KC; ++kc)
                        for (int nt = 0; nt < pc.nt; ++nt)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int cout = nt * 16 + (lane & 15);
                                    const int k = kc * 32 + (lane >> 4) * 8 + j;
                                    const int tapi = k / cg, cin = st * cg + k % cg;
                                    float val = 0.f;
                                    if (pair) {
                                        // row half h = pixel x + 2h: the record's pixels are filter columns (jx - h, jx + 1 - h) for it
                                        const int h = (lane & 15) >> 3;
                                        if (tapi < (int)taps.size() && (cin & 3) != 3) {
                                            Tap u = taps[tapi];
                                            u.kx = taps[tapi].kx + (cin >= 4 ? 1 : 0) - h;
                                            if (u.kx >= 0 && u.kx <= 8) val = (float)wval(cout & 7, cin & 3, u);
                                        }
                                    } else
                                    if (cout < L.cout && tapi < (int)taps.size()) val = (float)wval_t(cout, cin, taps[tapi]);
                                    uint16_t hi, lo;
                                    host_split(prec, val, hi, lo);
                                    const size_t base = ((((size_t)st * KC + kc) * pc.nt + nt) * parts) * 512 + (size_t)lane * 8 + j;
                                    wpk[base] = hi;
                                    if (parts == 2) wpk[base + 512] = lo;
                                }
                HIPCHK(hipMalloc((void **)&tp.tab[ps], tab.size() * sizeof(int)));
                HIPCHK(hipMemcpy(tp.tab[ps], tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
                HIPCHK(hipMalloc((void **)&tp.wpk[ps], wpk.size() * sizeof(uint16_t)));
                HIPCHK(hipMemcpy(tp.wpk[ps], wpk.data(), wpk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            }
            return DFFW_OK;
        };
        const TileCfg *cfg = tile_cfg_find(geo, pc.nt, cg, wide);
        if (cfg && cin_t % cg == 0) {
            const int rc = pack_tile(pc.tile, cfg, cg, false);
            if (rc != DFFW_OK) return rc;
        }
        // ... and on the 5 x 8 x 8 block where an instantiation exists (at most 4 output tiles per workgroup: wider layers always split there)
        if ((geo == G3S1 || geo == G3T) && cfg && cin_t % cg == 0 && L.cout >= 32 && !stem) {
            const TileCfg *ncfg = tile_cfg_find_shape(geo, std::min(pc.nt, 4), cg, 5, 8, 8);
            if (ncfg) {
                const int rc = pack_tile(pc.tile_narrow, ncfg, cg, false);
                if (rc != DFFW_OK) return rc;
            }
        }
        const TileCfg *pcfg = (stem && L.cout == 8 && !getenv("DFFW_NO_STEM_PAIR")) ? tile_cfg_find(G2P, 1, 8, wide) : nullptr;
        if (pcfg) {
            const int rc = pack_tile(pc.tile_pair, pcfg, 8, true);
            if (rc != DFFW_OK) return rc;
        }
    }
    // ---- third packing: conv_roll (rolling window along the slices) for the 16-channel 3x3x3 stride-1 layers ------
    // K order [dz][k5][32]: chunk k5 of a slice = in-slice taps 2*k5 and 2*k5+1 (tap 9 does not exist: zero weights),
    // lane group g -> tap 2*k5 + (g >> 1), channel octet g & 1
    if (geo == G3S1 && cin_pad == 16 && pc.nt == 1 && !stem) {
        pc.roll_pair = L.cout == 8;
        const int nch = pc.roll_pair ? ROLL_CHUNKS_PAIR : ROLL_CHUNKS;
        std::vector<uint16_t> wr((size_t)nch * parts * 512, 0);
        for (int c = 0; c < nch; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = lane & 15, gq = lane >> 4;
                    const int cin = (gq & 1) * 8 + j;
                    float val = 0.f;
                    if (!pc.roll_pair) {
                        const int dz = c / 5, k5 = c % 5;
                        const int tap9 = 2 * k5 + (gq >> 1);
                        if (row < L.cout && tap9 < 9) {
                            const int ky = tap9 / 3, kx = tap9 % 3;
                            val = (float)wval(row, cin, Tap{dz - 1, ky - 1, kx - 1, dz, ky, kx});
                        }
                    } else {
                        // pixel pairs: result rows 0-7 = channels of the even pixel, 8-15 = of the odd one; chunk
                        // (dz, ky, half) contracts input columns ix = 2*half + (gq >> 1) of the 4 the pair touches:
                        // the even pixel sees ix as filter column kx = ix, the odd pixel as kx = ix - 1
                        const int dz = c / 6, ky = (c % 6) / 2, half = c % 2;
                        const int ix = 2 * half + (gq >> 1);
                        const int cout = row & 7, kx = ix - (row >> 3);
                        if (cout < L.cout && kx >= 0 && kx <= 2) val = (float)wval(cout, cin, Tap{dz - 1, ky - 1, kx - 1, dz, ky, kx});
                    }
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wroll, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wroll, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_rollx_k2 (dffw_conv_rollx.hip): 3x3x3 stride 1, 32 -> 16 channels (`dres3.conv0`): conv_roll's plain chunk order per 16-channel
    // input half: [half][dz][k5], K octet g = (tap 2*k5 + (g >> 1), channel half*16 + (g & 1)*8 ..)
    if (geo == G3S1 && cin_pad == 32 && L.cin == 32 && L.cout == 16 && !stem && !shortcut_w && prec == P_BF16X3) {
        std::vector<uint16_t> wr((size_t)2 * ROLL_CHUNKS * parts * 512, 0);
        for (int half = 0; half < 2; ++half)
            for (int c = 0; c < ROLL_CHUNKS; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, gq = lane >> 4;
                        const int cin = half * 16 + (gq & 1) * 8 + j;
                        const int dz = c / 5, k5 = c % 5, tap9 = 2 * k5 + (gq >> 1);
                        float val = 0.f;
                        if (row < L.cout && tap9 < 9) {
                            const int ky = tap9 / 3, kx = tap9 % 3;
                            val = (float)wval(row, cin, Tap{dz - 1, ky - 1, kx - 1, dz, ky, kx});
                        }
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)half * ROLL_CHUNKS + c) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wroll_k2, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wroll_k2, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_slice32 (dffw_conv_slice.hip): per-slice 1x3x3, 32 -> 32 channels: chunk c = filter tap c ([ky][kx] order) x 32 channels (K octet g = channels 8g ..)
    if (geo == G2S1 && cin_pad == 32 && L.cin == 32 && L.cout == 32 && !shortcut_w && prec == P_BF16X3) {
        std::vector<uint16_t> wr((size_t)SLICE32_CHUNKS * 2 * parts * 512, 0);
        for (int c = 0; c < SLICE32_CHUNKS; ++c)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, gq = lane >> 4, ky = c / 3, kx = c % 3;
                        const float val = (float)wval(nt * 16 + row, gq * 8 + j, Tap{0, ky - 1, kx - 1, 0, ky, kx});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)c * 2 + nt) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wslice32, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wslice32, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_slice64 (dffw_conv_slice.hip): per-slice 1x3x3, 64 -> 64 channels: wave share = one 16-channel output tile; chunk c = (filter tap c / 2, channel half c % 2),
    // K octet g = channels 32 (c % 2) + 8g ..
    if (geo == G2S1 && cin_pad == 64 && L.cin == 64 && L.cout == 64 && !shortcut_w && prec == P_BF16X3) {
        std::vector<uint16_t> wr((size_t)4 * SLICE64_CHUNKS * parts * 512, 0);
        for (int nt = 0; nt < 4; ++nt)
            for (int c = 0; c < SLICE64_CHUNKS; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, gq = lane >> 4, tap = c / 2, ky = tap / 3, kx = tap % 3;
                        const float val = (float)wval(nt * 16 + row, (c % 2) * 32 + gq * 8 + j, Tap{0, ky - 1, kx - 1, 0, ky, kx});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)nt * SLICE64_CHUNKS + c) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wslice64, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wslice64, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ... and its HEAD variant for the level-3 alignment head's first conv over [features 32 | flow 2 | pad 6] records: chunk c < 9 = tap c x the 32 feature channels;
    // chunk 9 + k: K octet g = the record's fifth channel octet (flow_x, flow_y, zeros) at tap 4k + g (taps 9 .. 11: zero weights)
    if (geo == G2S1 && cin_pad == 40 && L.cin == 34 && L.cout == 64 && !shortcut_w && prec == P_BF16X3) {
        std::vector<uint16_t> wr((size_t)4 * SLICE64_HEAD_CHUNKS * parts * 512, 0);
        for (int nt = 0; nt < 4; ++nt)
            for (int c = 0; c < SLICE64_HEAD_CHUNKS; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, gq = lane >> 4;
                        const int tap = c < 9 ? c : 4 * (c - 9) + gq, cin = c < 9 ? gq * 8 + j : 32 + j;
                        float val = 0.f;
                        if (tap < 9 && cin < L.cin) val = (float)wval(nt * 16 + row, cin, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)nt * SLICE64_HEAD_CHUNKS + c) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wslice64, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wslice64, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ... and its CAT variant (conv_slice32_cat): 32 -> 32 over t with the block's folded 1x1x1 shortcut over x: chunks 0-8 = the taps over t, chunk 9 = the centre tap
    // over the shortcut's 32 channels (channels 32 .. 63 of the virtual concat)
    if (geo == G2S1 && cin_own == 32 && shortcut_w && shortcut_cin == 32 && L.cout == 32 && prec == P_BF16X3) {
        std::vector<uint16_t> wr((size_t)2 * SLICE32_CAT_CHUNKS * parts * 512, 0);
        for (int nt = 0; nt < 2; ++nt)
            for (int c = 0; c < SLICE32_CAT_CHUNKS; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, gq = lane >> 4, tap = c < 9 ? c : 4;
                        const float val = (float)wval(nt * 16 + row, (c < 9 ? 0 : 32) + gq * 8 + j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)nt * SLICE32_CAT_CHUNKS + c) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wslice64, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wslice64, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        pc.slice_cat = true;
    }
    // ---- conv_rollk (dffw_conv_rollk.hip): 3x3x3 stride 1, 32 / 64 -> 32 / 64 channels, the contraction split over the workgroup's waves: wave w =
    // (16-channel group w >> 1, tap half w & 1); tap slot s of a half = filter tap 14 * (w & 1) + s in [dz][ky][kx] order (tap 27: zero weights);
    // chunk c = slots 2c, 2c + 1; K octet g = (slot 2c + (g >> 1), channel octet g & 1 of the group)
    if (geo == G3S1 && (cin_pad == 32 || cin_pad == 64) && L.cin == cin_pad && L.cout % 32 == 0 && L.cout <= 64 && !stem && !shortcut_w && prec == P_BF16X3) {
        const int nw = cin_pad / 8, npair = L.cout / 32;
        std::vector<uint16_t> wr((size_t)npair * nw * ROLLK_CHUNKS * 2 * parts * 512, 0);
        for (int op = 0; op < npair; ++op)
            for (int wv = 0; wv < nw; ++wv)
                for (int c = 0; c < ROLLK_CHUNKS; ++c)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int row = lane & 15, gq = lane >> 4;
                                const int tap = (wv & 1) * 2 * ROLLK_CHUNKS + 2 * c + (gq >> 1);
                                const int cin = (wv >> 1) * 16 + (gq & 1) * 8 + j;
                                float val = 0.f;
                                if (tap < 27) {
                                    const int dz = tap / 9, ky = (tap % 9) / 3, kx = tap % 3;
                                    val = (float)wval((op * 2 + nt) * 16 + row, cin, Tap{dz - 1, ky - 1, kx - 1, dz, ky, kx});
                                }
                                uint16_t hi, lo;
                                host_split(prec, val, hi, lo);
                                const size_t base = (((((size_t)op * nw + wv) * ROLLK_CHUNKS + c) * 2 + nt) * parts) * 512 + (size_t)lane * 8 + j;
                                wr[base] = hi;
                                wr[base + 512] = lo;
                            }
        HIPCHK(hipMalloc((void **)&pc.wrollk, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wrollk, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_rollt (dffw_conv_rollt.hip): transposed 3x3x3 s(1,2,2), 32 / 64 -> 32 / 64 channels, the filter split over the workgroup's waves by output phase
    // and 16-channel output tile (rollt::Prog<role>: the wave's operand fragment sets and the accumulator slots = output phases each feeds).  A weight unit =
    // one tap x 32 channels (K octet g = channels 32 chunk + 8 g ..) x 16 outputs; units in the wave's set order, one per fed slot
    if (geo == G3T && (cin_pad == 32 || cin_pad == 64 || (cin_pad == 16 && L.cout == 16)) && L.cin == cin_pad &&
        ((L.cout % 32 == 0 && L.cout <= 64) || (L.cout == 16 && cin_pad <= 32)) && !shortcut_w && prec == P_BF16X3) {
        // (16 output channels, the wide form: the shares of the roles A32 / C32 once -- "waves" 0, 1 of one "half")
        const int nw = L.cout == 16 ? 2 : cin_pad / 8, nhalf = L.cout == 16 ? 1 : L.cout / 32;
        std::vector<uint16_t> wr((size_t)nhalf * nw * rollt::MAXU * parts * 512, 0);
        auto pack_role = [&](auto ROLE_, int oh, int wv) {
            using PR = rollt::Prog<decltype(ROLE_)::value>;
            const int cout0 = (oh * 2 + ((wv >> 1) & 1)) * 16;
            for (int i = 0; i < PR::NS; ++i) {
                int u = PR::ubase(i);
                for (int sl = 0; sl < PR::NACC; ++sl) {
                    if (!((PR::feeds(i) >> sl) & 1)) continue;
                    const int ph = PR::phase(sl), py = ph >> 1, px = ph & 1, d = PR::d(i), dy = PR::dy(i), dx = PR::dx(i);
                    const Tap tp{d - 1, dy, dx, 2 - d, rollt::tap_of(py, dy), rollt::tap_of(px, dx)};
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const float val = (float)wval(cout0 + (lane & 15), PR::chunk(i) * 32 + (lane >> 4) * 8 + j, tp);
                            uint16_t hi, lo;
                            host_split(prec, val, hi, lo);
                            const size_t base = ((((size_t)oh * nw + wv) * rollt::MAXU + u) * parts) * 512 + (size_t)lane * 8 + j;
                            wr[base] = hi;
                            wr[base + 512] = lo;
                        }
                    ++u;
                }
            }
        };
        for (int oh = 0; oh < nhalf; ++oh)
            for (int wv = 0; wv < nw; ++wv)
                switch (rollt_role(cin_pad == 16 ? 32 : cin_pad, wv)) {
                    case rollt::R_A: pack_role(std::integral_constant<int, rollt::R_A>{}, oh, wv); break;
                    case rollt::R_B: pack_role(std::integral_constant<int, rollt::R_B>{}, oh, wv); break;
                    case rollt::R_C: pack_role(std::integral_constant<int, rollt::R_C>{}, oh, wv); break;
                    case rollt::R_D: pack_role(std::integral_constant<int, rollt::R_D>{}, oh, wv); break;
                    case rollt::R_A32: pack_role(std::integral_constant<int, rollt::R_A32>{}, oh, wv); break;
                    default: pack_role(std::integral_constant<int, rollt::R_C32>{}, oh, wv); break;
                }
        HIPCHK(hipMalloc((void **)&pc.wrollt, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wrollt, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- srd_roll stage C: the attention convs of the 8-channel SRD block (DEN.py:322-323) in pixel-pair form.  Result row
    // m = (pixel m >> 3 of the pair, channel m & 7).  3x1x1: chunk 0 K octet g = (pixel g >> 1, slice g & 1), chunk 1 = slice 2 in the
    // 1x1x1 form (its operand is what stage B of the same step leaves in registers).  1x1x1: K octet g = (pixel g >> 1, input channels 4*(g & 1)..+3 as [hi x4 | lo x4] of the split
    // operand): fragment 0 carries w_hi against both halves (w_hi*a_hi + w_hi*a_lo), fragment 1 w_lo against the hi half.
    if (!L.transposed && L.kh == 1 && L.kw == 1 && L.cin == 8 && L.cout == 8 && !bn && !conv_bias) {
        const int nfrag = L.kd == 3 ? 2 * parts : parts;
        std::vector<uint16_t> wr((size_t)nfrag * 512, 0);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int row = lane & 15, gq = lane >> 4, px = row >> 3, co = row & 7;
                const bool mine = (gq >> 1) == px;
                if (L.kd == 3) {
                    {
                        const float val = mine ? (float)wval(co, j, Tap{0, 0, 0, gq & 1, 0, 0}) : 0.f;
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        wr[(size_t)lane * 8 + j] = hi;
                        if (parts == 2) wr[(size_t)512 + lane * 8 + j] = lo;
                    }
                    {   // chunk 1 = slice tap 2 against the operand stage B leaves in registers (the 1x1x1 form below)
                        const float val = mine ? (float)wval(co, 4 * (gq & 1) + (j & 3), Tap{0, 0, 0, 2, 0, 0}) : 0.f;
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        wr[((size_t)parts) * 512 + lane * 8 + j] = hi;
                        if (parts == 2 && j < 4) wr[((size_t)parts + 1) * 512 + lane * 8 + j] = lo;
                    }
                } else if (L.kd == 1) {
                    const int ci = 4 * (gq & 1) + (j & 3);
                    const float val = mine ? (float)wval(co, ci, Tap{0, 0, 0, 0, 0, 0}) : 0.f;
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    wr[(size_t)lane * 8 + j] = hi;
                    if (parts == 2 && j < 4) wr[512 + (size_t)lane * 8 + j] = lo;
                }
            }
        if (L.kd == 3 || L.kd == 1) {
            HIPCHK(hipMalloc((void **)&pc.watt, wr.size() * sizeof(uint16_t)));
            HIPCHK(hipMemcpy(pc.watt, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    }
    // the same for the 16-channel block (srd_roll16, no pixel pairs: result row = channel).  3x1x1: chunk 0 K octet g =
    // (slice g >> 1, channel octet g & 1), chunk 1 = slice 2 in the 1x1x1 form.  1x1x1: K octet g = input channels 4g..4g+3 as
    // [hi x4 | lo x4]; fragment 0 = w_hi against both halves, fragment 1 = w_lo against the hi half.
    if (!L.transposed && L.kh == 1 && L.kw == 1 && L.cin == 16 && L.cout == 16 && !bn && !conv_bias && (L.kd == 3 || L.kd == 1)) {
        const int nfrag = L.kd == 3 ? 2 * parts : parts;
        std::vector<uint16_t> wr((size_t)nfrag * 512, 0);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int co = lane & 15, gq = lane >> 4;
                if (L.kd == 3) {
                    {
                        const float val = (float)wval(co, (gq & 1) * 8 + j, Tap{0, 0, 0, gq >> 1, 0, 0});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        wr[(size_t)lane * 8 + j] = hi;
                        if (parts == 2) wr[(size_t)512 + lane * 8 + j] = lo;
                    }
                    {   // chunk 1 = slice tap 2 against the operand stage B leaves in registers (the 1x1x1 form below)
                        const float val = (float)wval(co, 4 * gq + (j & 3), Tap{0, 0, 0, 2, 0, 0});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        wr[((size_t)parts) * 512 + lane * 8 + j] = hi;
                        if (parts == 2 && j < 4) wr[((size_t)parts + 1) * 512 + lane * 8 + j] = lo;
                    }
                } else {
                    const float val = (float)wval(co, 4 * gq + (j & 3), Tap{0, 0, 0, 0, 0, 0});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    wr[(size_t)lane * 8 + j] = hi;
                    if (parts == 2 && j < 4) wr[512 + (size_t)lane * 8 + j] = lo;
                }
            }
        HIPCHK(hipMalloc((void **)&pc.watt, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.watt, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // the same for the 32-channel block (srd_attention_mfma, two 16-channel output tiles).  3x1x1: chunk k = slice k, K octet g =
    // input channels 8g..8g+7.  1x1x1: channel chunk c = input channels 16c..16c+15, K octet g = channels 16c+4g..+3 as [hi | lo].
    if (!L.transposed && L.kh == 1 && L.kw == 1 && L.cin == 32 && L.cout == 32 && !bn && !conv_bias && (L.kd == 3 || L.kd == 1)) {
        const int nfrag = L.kd == 3 ? 3 * 2 * parts : 2 * parts * 2;
        std::vector<uint16_t> wr((size_t)nfrag * 512, 0);
        for (int nt = 0; nt < 2; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = nt * 16 + (lane & 15), gq = lane >> 4;
                    if (L.kd == 3) {
                        for (int k = 0; k < 3; ++k) {
                            const float val = (float)wval(co, gq * 8 + j, Tap{0, 0, 0, k, 0, 0});
                            uint16_t hi, lo;
                            host_split(prec, val, hi, lo);
                            wr[((size_t)(k * 2 + nt) * parts) * 512 + lane * 8 + j] = hi;
                            if (parts == 2) wr[((size_t)(k * 2 + nt) * parts + 1) * 512 + lane * 8 + j] = lo;
                        }
                    } else {
                        for (int c = 0; c < 2; ++c) {
                            const float val = (float)wval(co, 16 * c + 4 * gq + (j & 3), Tap{0, 0, 0, 0, 0, 0});
                            uint16_t hi, lo;
                            host_split(prec, val, hi, lo);
                            wr[((size_t)(c * parts) * 2 + nt) * 512 + lane * 8 + j] = hi;
                            if (parts == 2 && j < 4) wr[((size_t)(c * parts + 1) * 2 + nt) * 512 + lane * 8 + j] = lo;
                        }
                    }
                }
        HIPCHK(hipMalloc((void **)&pc.watt, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.watt, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- of_roll (alignment network, stride-1 blocks): conv.0 with 8 (3 real) input channels -> 16: chunk k, K octet g = tap 4k + g;
    // conv.2 16 -> 16 with the block's 1x1x1 shortcut folded in (shortcut_w): 5 chunks over t as below + ONE chunk whose K octet g
    // = channel octet g of the block input at the centre tap
    if (geo == G2S1 && cin_pad == 8 && L.cout == 16 && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)3 * parts * 512, 0);
        for (int c = 0; c < 3; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int tap = 4 * c + (lane >> 4);
                    float val = 0.f;
                    if (tap < 9) val = (float)wval(lane & 15, j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // of_roll8: conv.2 8 -> 8 with the folded shortcut, pixel-pair form: chunk ky as for srd_roll (K octet g = input column 2*pair + g),
    // chunk 3 = shortcut: K octet g < 2 = the 8 block-input channels of pixel 2*pair + g, seen only by that pixel's result rows
    if (geo == G2S1 && cin_own == 8 && L.cout == 8 && shortcut_w && shortcut_cin <= 8) {
        std::vector<uint16_t> wr((size_t)4 * parts * 512, 0);
        for (int c = 0; c < 4; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = lane & 15, gq = lane >> 4, cout = row & 7, px = row >> 3;
                    float val = 0.f;
                    if (c < 3) {
                        const int kx = gq - px;
                        if (kx >= 0 && kx <= 2) val = (float)wval(cout, j, Tap{0, c - 1, kx - 1, 0, c, kx});
                    } else if (gq == px && j < shortcut_cin) {
                        val = (float)wval(cout, cin_own + j, Tap{0, 0, 0, 0, 1, 1});
                    }
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (geo == G2S1 && cin_own == 16 && L.cout == 16 && shortcut_w && (shortcut_cin + 7) / 8 * 8 <= 16) {
        std::vector<uint16_t> wr((size_t)OF_CHUNKS_B * parts * 512, 0);
        for (int c = 0; c < OF_CHUNKS_B; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = lane & 15, gq = lane >> 4;
                    float val = 0.f;
                    if (c < 5) {
                        const int tap = 2 * c + (gq >> 1);
                        if (tap < 9) val = (float)wval(co, (gq & 1) * 8 + j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                    } else if (gq * 8 + j < shortcut_cin) {
                        val = (float)wval(co, cin_own + gq * 8 + j, Tap{0, 0, 0, 0, 1, 1});
                    }
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- head_warp<CF = 16>: the [cur (16) | flow (2)] part of the level-2 alignment head's first conv, 32 output channels: chunk k,
    // K octet g = o = 4k + g -> (filter tap o / 3, channel octet o % 3) of the 24-channel records the kernel builds in LDS
    if (geo == G2S1 && L.cin == 18 && cin_pad == 24 && L.cout == 32 && !shortcut_w) {
        constexpr int NCHW = head_warp_chunks(16);
        std::vector<uint16_t> wr((size_t)NCHW * 2 * parts * 512, 0);
        for (int c = 0; c < NCHW; ++c)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nt * 16 + (lane & 15), o = 4 * c + (lane >> 4), tap = o / 3, oct = o % 3;
                        float val = 0.f;
                        if (o < 27) val = (float)wval(co, oct * 8 + j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = (((size_t)c * 2 + nt) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        if (parts == 2) wr[base + 512] = lo;
                    }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- of_s2 (down-sampling block 8 -> 16 of the alignment network): conv.0 = 1x3x3 stride (1,2,2), 8 -> 16: chunk k, K octet g = filter
    // tap 4k + g; its 1x1x1 stride-2 shortcut (no BatchNorm, no bias): one chunk, K octet 0 = the 8 input channels
    if (geo == G2S2 && cin_pad == 8 && L.cout == 16 && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)3 * parts * 512, 0);
        for (int c = 0; c < 3; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = lane & 15, tap = 4 * c + (lane >> 4);
                    float val = 0.f;
                    if (tap < 9) val = (float)wval(co, j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (!L.transposed && L.kd == 1 && L.kh == 1 && L.kw == 1 && L.sh == 2 && cin_pad == 8 && L.cout == 16 && !bn && !conv_bias && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)parts * 512, 0);
        for (int lane = 0; lane < 16; ++lane)        // K octet 0 only
            for (int j = 0; j < 8; ++j) {
                uint16_t hi, lo;
                host_split(prec, (float)wval(lane, j, Tap{0, 0, 0, 0, 0, 0}), hi, lo);
                wr[(size_t)lane * 8 + j] = hi;
                if (parts == 2) wr[512 + (size_t)lane * 8 + j] = lo;
            }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- srd_roll16: the per-slice 1x3x3 16 -> 16 convs: chunk k, K octet g = (filter tap 2k + (g >> 1), channel octet g & 1)
    // (packed with a sixth, all-zero chunk: the same buffer then serves as the second conv of of_roll_kernel, whose shortcut chunk
    // it leaves empty, for plain conv -> conv chains such as the alignment heads' .2.0 -> .4.0)
    if (geo == G2S1 && cin_pad == 16 && L.cout == 16 && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)OF_CHUNKS_B * parts * 512, 0);
        for (int c = 0; c < SRD16_CHUNKS; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = lane & 15, gq = lane >> 4, tap = 2 * c + (gq >> 1);
                    float val = 0.f;
                    if (tap < 9) val = (float)wval(co, (gq & 1) * 8 + j, Tap{0, tap / 3 - 1, tap % 3 - 1, 0, tap / 3, tap % 3});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- srd_roll: the per-slice 1x3x3 8 -> 8 convs of the fused SRD block, in pixel-pair form: chunk = filter row ky;
    // result rows 0-7 = channels of the even pixel of a pair, rows 8-15 = of the odd one; K octet g = input column 2*pair + g,
    // which the even pixel sees as filter column g and the odd pixel as filter column g - 1
    if (geo == G2S1 && cin_pad == 8 && L.cout == 8 && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)SRD_CHUNKS * parts * 512, 0);
        for (int c = 0; c < SRD_CHUNKS; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = lane & 15, gq = lane >> 4;
                    const int cout = row & 7, kx = gq - (row >> 3);
                    float val = 0.f;
                    if (kx >= 0 && kx <= 2) val = (float)wval(cout, j, Tap{0, c - 1, kx - 1, 0, c, kx});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wsrd, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wsrd, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_roll_efd: 3x3x3 8 -> 16 (stride 1 on the pooled volume, or stride (1,2,2)): chunk (dz, k3), K octet g = tap 4*k3 + g
    if ((geo == G3S1 || geo == G3S2) && cin_pad == 8 && L.cout == 16 && !shortcut_w) {
        std::vector<uint16_t> wr((size_t)ROLL_CHUNKS_8 * parts * 512, 0);
        for (int c = 0; c < ROLL_CHUNKS_8; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = lane & 15, tap = 4 * (c % 3) + (lane >> 4), dz = c / 3;
                    float val = 0.f;
                    if (tap < 9) val = (float)wval(co, j, Tap{dz - 1, tap / 3 - 1, tap % 3 - 1, dz, tap / 3, tap % 3});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wroll8, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wroll8, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_roll_s2: 3x3x3 stride (1,2,2), 16 / 32 -> 16 / 32 / 64 channels: per (16-channel output tile, 16-channel input half) 15 chunks
    // [dz][k5], K octet g = (in-slice tap 2*k5 + (g >> 1), channel octet g & 1 of the half), as conv_roll's plain form
    // (the same order for the stride-1 16 -> 32 layer `FM_conv2.0.max_pooling.1`: the pooled branch of conv_efd16)
    const bool pool15 = geo == G3S1 && cin_pad == 16 && L.cout == 32 && !shortcut_w && !stem;
    if ((geo == G3S2 && (cin_pad == 16 || cin_pad == 32) && L.cout % 16 == 0 && L.cout <= 64 && !(cin_pad == 16 && L.cout == 64) && !shortcut_w) || pool15) {
        const int ntl = L.cout / 16, khn = cin_pad / 16;
        std::vector<uint16_t> wr((size_t)ntl * khn * ROLL_CHUNKS * parts * 512, 0);
        for (int nt = 0; nt < ntl; ++nt)
            for (int kh = 0; kh < khn; ++kh)
                for (int c = 0; c < ROLL_CHUNKS; ++c)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int row = lane & 15, gq = lane >> 4;
                            const int dz = c / 5, k5 = c % 5, tap9 = 2 * k5 + (gq >> 1);
                            float val = 0.f;
                            if (tap9 < 9) val = (float)wval(nt * 16 + row, kh * 16 + (gq & 1) * 8 + j, Tap{dz - 1, tap9 / 3 - 1, tap9 % 3 - 1, dz, tap9 / 3, tap9 % 3});
                            uint16_t hi, lo;
                            host_split(prec, val, hi, lo);
                            const size_t base = ((((size_t)nt * khn + kh) * ROLL_CHUNKS + c) * parts) * 512 + (size_t)lane * 8 + j;
                            wr[base] = hi;
                            if (parts == 2) wr[base + 512] = lo;
                        }
        uint16_t **dst = pool15 ? &pc.wroll15 : &pc.wroll_s2;
        HIPCHK(hipMalloc((void **)dst, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(*dst, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_roll_t32: transposed 3x3x3 s(1,2,2), 32 -> 16 channels, one fragment set per output row phase py.  A chunk = one
    // tap x 32 channels (K octet g = channel octet g).  Enumeration (must match the kernel): x phase 0 first: (window slice d,
    // row tap rt) with filter column 1 at input column x; then x phase 1: (d, rt, ct): ct = 0 -> filter column 2 at x, ct = 1 ->
    // filter column 0 at x+1.  Row taps: py = 0: filter row 1 at input row y; py = 1: rt = 0 -> filter row 2 at y, rt = 1 -> row 0 at y+1.
    if (geo == G3T && (cin_pad == 32 || cin_pad == 16) && L.cout == 16) {   // (16 input channels: octets 2, 3 get zero weights)
        std::vector<uint16_t> wr((size_t)(ROLL_CHUNKS_T32_0 + ROLL_CHUNKS_T32_1) * parts * 512, 0);
        size_t chunk0 = 0;
        for (int py = 0; py < 2; ++py) {
            const int nrow = py ? 2 : 1, nch0 = 3 * nrow, nch = 9 * nrow;
            for (int c = 0; c < nch; ++c) {
                const int e = c - nch0;
                const int d = c < nch0 ? c / nrow : e / (2 * nrow);
                const int rt = c < nch0 ? c % nrow : (e / 2) % nrow;
                const int ct = c < nch0 ? 0 : e % 2;
                const int ky = py ? (rt == 0 ? 2 : 0) : 1, dy = (py && rt == 1) ? 1 : 0;
                const int kx = c < nch0 ? 1 : (ct == 0 ? 2 : 0), dx = ct;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int cin = (lane >> 4) * 8 + j;
                        const float val = cin < cin_pad ? (float)wval(lane & 15, cin, Tap{d - 1, dy, dx, 2 - d, ky, kx}) : 0.f;
                        uint16_t hi, lo;
                        host_split(prec, val, hi, lo);
                        const size_t base = ((chunk0 + c) * parts) * 512 + (size_t)lane * 8 + j;
                        wr[base] = hi;
                        if (parts == 2) wr[base + 512] = lo;
                    }
            }
            chunk0 += nch;
        }
        HIPCHK(hipMalloc((void **)&pc.wroll_t32, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wroll_t32, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    // ---- conv_roll_t: transposed 3x3x3 s(1,2,2), 16 -> 8 channels.  Result rows 0-7 = output pixel 2x, rows 8-15 = pixel
    // 2x+1; chunk c < 3: output row phase py = 0 (filter row 1 at input row y), slice c of the window; c >= 3: py = 1,
    // slice (c-3)/2, filter row 2 at input row y ((c-3) even) or filter row 0 at input row y+1 (odd).  Lane group g
    // contracts input column x + (g >> 1), channel octet g & 1: pixel 2x sees only column x (filter column 1), pixel 2x+1
    // sees column x (filter column 2) and column x+1 (filter column 0).  Window slice d is input slice oz-1+d = filter slice 2-d.
    if (geo == G3T && cin_pad == 16 && L.cout == 8) {
        std::vector<uint16_t> wr((size_t)ROLL_CHUNKS_T * parts * 512, 0);
        for (int c = 0; c < ROLL_CHUNKS_T; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = lane & 15, gq = lane >> 4;
                    const int cout = row & 7, px = row >> 3, dx = gq >> 1, cin = (gq & 1) * 8 + j;
                    const int d = c < 3 ? c : (c - 3) / 2;
                    const bool down = c >= 3 && ((c - 3) & 1);
                    const int ky = c < 3 ? 1 : (down ? 0 : 2), dy = down ? 1 : 0;
                    int kx = -1;
                    if (px == 0 && dx == 0) kx = 1;
                    if (px == 1) kx = dx == 0 ? 2 : 0;
                    float val = 0.f;
                    if (kx >= 0) val = (float)wval(cout, cin, Tap{d - 1, dy, dx, 2 - d, ky, kx});
                    uint16_t hi, lo;
                    host_split(prec, val, hi, lo);
                    const size_t base = ((size_t)c * parts) * 512 + (size_t)lane * 8 + j;
                    wr[base] = hi;
                    if (parts == 2) wr[base + 512] = lo;
                }
        HIPCHK(hipMalloc((void **)&pc.wroll_t, wr.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(pc.wroll_t, wr.data(), wr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    return DFFW_OK;
}

// ---- workspace arena ---------------------------------------------------------------------------
// Offsets into the caller's workspace; first-fit with coalescing.  The graph is static for a given
// (B,N,H,W), so a dry run of the same code computes the exact peak (dffw_workspace_bytes).
class Arena {
  public:
    explicit Arena(int64_t cap = INT64_MAX) { free_.push_back({0, cap}); }
    int64_t alloc(int64_t bytes) {
        bytes = (bytes + 255) & ~(int64_t)255;
        for (size_t i = 0; i < free_.size(); ++i) {
            if (free_[i].second >= bytes) {
                const int64_t off = free_[i].first;
                free_[i].first += bytes;
                free_[i].second -= bytes;
                if (free_[i].second == 0) free_.erase(free_.begin() + i);
                live_[off] = bytes;
                peak_ = std::max(peak_, off + bytes);
                return off;
            }
        }
        return -1;
    }
    void release(int64_t off) {
        auto it = live_.find(off);
        if (it == live_.end()) return;
        std::pair<int64_t, int64_t> blk{off, it->second};
        live_.erase(it);
        auto pos = std::lower_bound(free_.begin(), free_.end(), blk);
        pos = free_.insert(pos, blk);
        size_t i = pos - free_.begin();
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
            free_[i].second += free_[i + 1].second;
            free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
            free_[i - 1].second += free_[i].second;
            free_.erase(free_.begin() + i);
        }
    }
    int64_t peak() const { return peak_; }

  private:
    int64_t peak_ = 0;
    std::vector<std::pair<int64_t, int64_t>> free_;
    std::map<int64_t, int64_t> live_;
};

}  // namespace dffw

using namespace dffw;

extern "C" char **environ;

// ---- engine ------------------------------------------------------------------------------------
struct ProfRec {
    std::string kernel, layer;
    double flops = 0, bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

struct dffw_engine {
    int device = 0, net = 0, prec = 0;
    std::map<std::string, PackedConv> convs;
    bool profiling = false;
    std::vector<ProfRec> recs;
    uint16_t *zero_page = nullptr;  // 256 zero bytes: what out-of-volume LDS-DMA lanes read
    // side streams + events for the small-shape regime, where single launches cannot fill the chip and the
    // independent branches of the graph (pyramid scales, regression heads) run next to the main chain
    static constexpr int NSIDE = 2, NEV = 16;
    hipStream_t side[NSIDE] = {nullptr, nullptr};
    hipEvent_t ev[NEV] = {};
    int ensure_side() {
        if (side[0]) return DFFW_OK;
        for (int i = 0; i < NSIDE; ++i) HIPCHK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
        for (int i = 0; i < NEV; ++i) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        return DFFW_OK;
    }
    int ensure_zero_page() {
        if (zero_page) return DFFW_OK;
        HIPCHK(hipMalloc((void **)&zero_page, 256));
        HIPCHK(hipMemset(zero_page, 0, 256));
        return DFFW_OK;
    }
    void clear_recs() {
        for (auto &r : recs) {
            if (r.e0) (void)hipEventDestroy(r.e0);
            if (r.e1) (void)hipEventDestroy(r.e1);
        }
        recs.clear();
    }
    ~dffw_engine() {
        clear_recs();
        if (zero_page) (void)hipFree(zero_page);
        for (int i = 0; i < NSIDE; ++i)
            if (side[i]) (void)hipStreamDestroy(side[i]);
        for (int i = 0; i < NEV; ++i)
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        for (auto &kv : convs) free_packed(kv.second);
    }
};

namespace dffw {

static bool getenv_flag(const char *name) {
    const char *v = getenv(name);
    return v && *v && *v != '0';
}

// The DFFW_* kernel-path switches, read from the environment ONCE per forward (Run's constructor): the
// graph code below never calls getenv itself, so a forward sees one consistent set and pays for one scan of environ.
#define DFFW_SWITCHES(X)                                                                                                 \
    X(NO_CONCURRENT) X(NO_CONF_FORK) X(NO_FUSED_ATTENTION) X(NO_FUSED_EFD) X(NO_FUSED_OF) X(NO_FUSED_POOL) X(NO_FUSED_SRD) \
    X(NO_FUSED_STEM) X(NO_HEAD_SPLIT) X(NO_ROLL) X(NO_ROLL_S2) X(NO_SPLIT) X(NO_SPLITK)  \
    X(NO_STEM_PAIR) X(NO_TILE) X(NO_SMALL) X(NO_ROLL_S2_WIDE) X(NO_HEAD_SUMS) X(NO_HEAD_SUMS_FUSED) X(NO_HEAD_WARP) X(NO_OF_FIRST) \
    X(NO_LEAN_TILE) X(NO_LEAN_ROLL) X(NO_ROLLX) X(NO_ROLLK) X(NO_ROLLT) X(NO_TEAMS) X(NO_SLICE32) X(NO_REGRESS_FUSED) X(NO_STEM_PIPE) X(NO_POOL3) X(NO_NARROW)
enum SwitchId {
#define X_ID(n) SW_##n,
    DFFW_SWITCHES(X_ID)
#undef X_ID
    SW_COUNT
};
// A DFFW_* variable in the environment that nothing reads (a retired switch, a typo) is reported once per process: a retired switch that is silently ignored turns an
// A/B script into base against base.
static void warn_unknown_switches() {
    static bool done = false;
    if (done) return;
    done = true;
    static const char *const known[] = {
#define X_NM(n) "DFFW_" #n,
        DFFW_SWITCHES(X_NM)
#undef X_NM
        "DFFW_ROLL_WGS", "DFFW_SRD_WGS", "DFFW_SMALL_MAX_UNITS", "DFFW_ROLL_ZSPLIT", "DFFW_KSPLIT_TARGET", "DFFW_SPLIT_S64", "DFFW_SPLIT_T64", "DFFW_NARROW_MAX",
        "DFFW_WARM_MAX_WGS", "DFFW_ROLLK_MERGE_BELOW", "DFFW_ROLLT_MIN_UNITS", "DFFW_ROLL_MIN_UNITS", "DFFW_SPLIT_WG", "DFFW_DEBUG_FLAGS", "DFFW_CONCURRENT_MAX_PIXELS",
        "DFFW_SRD_PIPE", "DFFW_REDIR_SIDE", "DFFW_TEAM_MIN_WGS", "DFFW_TEAM_MAX_WGS", "DFFW_CONCURRENT_MIN_PIXELS", "DFFW_TRACE_LAYER", "DFFW_TRACE_OUT", "DFFW_NO_STEM_PAIR", "DFFW_RCCL_LIB", "DFFW_LIB_PATH", "DFFW_NO_PROBE", "DFFW_PRECISION",
        "DFFW_BENCH_ONE_DEVICE", "DFFW_BENCH_TIMEOUT_S",
        // development builds only (make ABL=1 / TRACE=1)
        "DFFW_SRD_ABL", "DFFW_ROLLX_ABL", "DFFW_ROLLX_NTS", "DFFW_ROLLK_ABL", "DFFW_ROLLK_SKEW"};
    for (char **e = ::environ; e && *e; ++e) {
        if (strncmp(*e, "DFFW_", 5) != 0) continue;
        const char *eq = strchr(*e, '=');
        const size_t n = eq ? (size_t)(eq - *e) : strlen(*e);
        bool ok = false;
        for (const char *k : known) ok = ok || (strlen(k) == n && strncmp(k, *e, n) == 0);
        if (!ok) fprintf(stderr, "libdffw: environment variable %.*s is not a switch this library reads (retired or misspelt?) -- ignored\n", (int)n, *e);
    }
}

struct Switches {
    bool f[SW_COUNT];
    int roll_wgs = 0, roll_zsplit = 0, srd_wgs = 0, split_wg = 256, debug_flags = 0, small_max_units = 0, roll_min_units = 192, rollt_min_units = 128, rollk_merge_below = 1 << 30, redir_side = -1, team_min_wgs = 0, team_max_wgs = 320, ksplit_target = 512, warm_max_wgs = 1024, narrow_max = 40, split_t64 = 1025, split_s64 = 1025;
    bool srd_pipe = false;   // DFFW_SRD_PIPE=1: the 16-channel SRD block on srd_pipe16 (one barrier per step) instead of srd_roll16
    int64_t concurrent_max_pixels = -1;   // < 0: no limit
    int64_t concurrent_min_pixels = 400000;   // below: one stream (a 5x224x224 stack, 0.25M: 0.799 -> 0.785 ms on one stream; 10x256x256, 0.66M: 0.915 -> 0.905 on three)
    const char *trace_layer = nullptr, *trace_out = nullptr;
    bool on(int id) const { return f[id]; }
    // the switches the kernel launchers consult, as ConvArgs::dbg bits (so that no launcher calls getenv)
    int path_bits() const;
    static Switches read() {
        Switches s;
        int i = 0;
#define X_RD(n) s.f[i++] = getenv_flag("DFFW_" #n);
        DFFW_SWITCHES(X_RD)
#undef X_RD
        auto geti = [](const char *name, int lo, int dflt) {
            const char *z = getenv(name);
            return (z && atoi(z) >= lo) ? atoi(z) : dflt;
        };
        s.roll_wgs = geti("DFFW_ROLL_WGS", 8, 0);
        s.srd_wgs = geti("DFFW_SRD_WGS", 8, 0);
        s.small_max_units = geti("DFFW_SMALL_MAX_UNITS", 0, 0);   // (256 measured 6 % faster on one 5x224x224 stack, level on 10x256x256 -- but a
                                                                  // batch-1 call then differs from the same stack inside a batch by 1.5e-5: off)
        s.roll_zsplit = geti("DFFW_ROLL_ZSPLIT", 1, 0);
        s.ksplit_target = geti("DFFW_KSPLIT_TARGET", 1, 512);
        s.redir_side = geti("DFFW_REDIR_SIDE", 0, -1);   // redir1 / redir2 on the side streams: 1 always, 0 never, unset: from 2M stack pixels
        s.team_min_wgs = geti("DFFW_TEAM_MIN_WGS", 0, 0);
        s.team_max_wgs = geti("DFFW_TEAM_MAX_WGS", 1, 320);
        s.split_s64 = geti("DFFW_SPLIT_S64", 1, 1025);   // ... and a stride-1 3x3x3 layer with 64 outputs (End_to_End dres16_* at batch 8: 0.093 -> 0.076 ms on two 2-tile workgroups per tile; 257 = round 4)
        s.split_t64 = geti("DFFW_SPLIT_T64", 1, 1025);   // tile count below which a transposed layer with 64 outputs splits its output channels (257 = round 4)
        s.narrow_max = geti("DFFW_NARROW_MAX", 1, 40);   // widest grid that may take the 5 x 8 x 8 block when its own block leaves the chip short of workgroups (8 = round 4)
        // conv_tile launches of at most this many (tile, channel-split) workgroups touch the weight lines of their whole contraction walk first
        // (TileArgs::warm; 0: never).  Measured r03 on 10x256x256 stacks, ms per forward at 0 / 256 / 1024 / always: batch 2 1.29 / 1.17 / 1.17 /
        // 1.17, batch 8 2.74 / 2.65 / 2.63 / 2.63, batch 16 4.71 / 4.66 / 4.62 / 4.66, batch 32 8.58 / 8.57 / 8.57 / 8.70
        s.warm_max_wgs = geti("DFFW_WARM_MAX_WGS", 0, 1024);
        // conv_rollk layers with 64 outputs on fewer columns than this run their two output halves as ONE launch (grid.y = 2); 1: never.  Round 6, same-run layer
        // tables at batch 32: the 16 x 16-grid layers (128 columns: one unit per CU) 0.058 -> 0.050-0.054 ms against conv_tile, dres0.0 0.099 -> 0.090, dres0.2 /
        // dres2.conv2 -1..2 % against two launches
        s.rollk_merge_below = geti("DFFW_ROLLK_MERGE_BELOW", 0, 1 << 30);
        s.rollt_min_units = geti("DFFW_ROLLT_MIN_UNITS", 1, 128);   // (column, output half) units the transposed streaming kernel conv_rollt needs in its 8-wave forms, 1.5x that in the 4-wave form (DFFW_ROLL_MIN_UNITS lowers it too)
        s.roll_min_units = geti("DFFW_ROLL_MIN_UNITS", 1, 192);   // columns a layer needs for its persistent streaming kernel (measured 16 ... 256
                                                                  // at batch 1 and 4: 192 is 3.6 % faster than 256 on one 5x224x224 stack, level elsewhere; <= 32 slower)
        s.rollt_min_units = std::min(s.rollt_min_units, s.roll_min_units);
        { const char *z = getenv("DFFW_SPLIT_WG"); s.split_wg = z ? atoi(z) : 256; }   // measured best of 64/128/256/512 at batch 1, 4, 8
        { const char *z = getenv("DFFW_DEBUG_FLAGS"); s.debug_flags = z ? atoi(z) : 0; }
        { const char *z = getenv("DFFW_CONCURRENT_MAX_PIXELS"); s.concurrent_max_pixels = z ? atoll(z) : -1; }
        { const char *z = getenv("DFFW_CONCURRENT_MIN_PIXELS"); if (z) s.concurrent_min_pixels = atoll(z); }
        s.srd_pipe = getenv_flag("DFFW_SRD_PIPE");
        warn_unknown_switches();
        s.trace_layer = getenv("DFFW_TRACE_LAYER");
        s.trace_out = getenv("DFFW_TRACE_OUT");
        return s;
    }
};

inline int Switches::path_bits() const {
    return (f[SW_NO_LEAN_TILE] ? DFFW_ARGS_NO_LEAN_TILE : 0) | (f[SW_NO_LEAN_ROLL] ? DFFW_ARGS_NO_LEAN_ROLL : 0) | (f[SW_NO_ROLLX] ? DFFW_ARGS_NO_ROLLX : 0) |
           (f[SW_NO_ROLLK] ? DFFW_ARGS_NO_ROLLK : 0) | (f[SW_NO_SLICE32] ? DFFW_ARGS_NO_SLICE32 : 0) | (f[SW_NO_ROLLT] ? DFFW_ARGS_NO_ROLLT : 0);
}

struct ConvOpt {
    const Act *in1 = nullptr;
    const Act *res0 = nullptr, *res1 = nullptr;
    bool res_bcast = false;  // res0 has one slice per sample and is added to every output slice
    int relu = 0;
    Act *out_pre = nullptr;  // receives the pre-residual value (allocated here)
    float *outf = nullptr;   // fp32 planar output (B,outf_ch,N,H,W) instead of an activation volume
    int outf_ch = 1;
    const float *fs32 = nullptr;  // stem: read the fp32 focal stack directly (in0 then only carries the geometry)
    bool raw = false;   // stem: fs32 is a device-side RawStack descriptor (raw uint8 / 0..255 stack, normalised and padded on the fly)
    float *sums = nullptr;      // per-slice 1x3x3 conv + ReLU whose result only feeds plane sums: nothing is stored, `sums` receives per output
                                // row segment of 16 pixels [sum | first pixel | last pixel][Cout] fp32 (conv_tile's row-sums variant;
                                // Run::sums_conv_ok() says when it exists; out is not allocated)
    const char *cls = nullptr;  // name of a 1x1x1 C->1 layer to apply to the final value inside the epilogue
    float *cls_out = nullptr;   // its fp32 score volume
    bool discard = false;       // the activation output itself is not needed (only cls_out / out_pre)
};

struct Run {
    dffw_engine *e;
    hipStream_t s;
    Arena arena;
    bool dry;
    char *ws;
    int err = DFFW_OK;
    const dffw_tap *taps = nullptr;
    int n_taps = 0;

    // ---- branch concurrency (small shapes only; decided from the shape alone so that the dry run that sizes the
    // workspace takes the same allocation path).  Inside a forked section buffers are not recycled: a block released
    // after a launch on one stream must not be handed to a launch on another stream that may run earlier.  Outside
    // it recycling continues as usual — measured: running the whole batch-1 forward without recycling costs 17 %
    // (the working set stops fitting the last-level cache). ----
    hipStream_t main_s;
    bool concurrent = false;
    bool forked = false;
    int ev_next = 0;
    unsigned side_open = 0;   // side streams forked and not yet joined
    std::vector<void *> deferred;

    const Switches sw;   // the DFFW_* switches as they were when this forward started

    Run(dffw_engine *e_, hipStream_t s_, bool dry_, char *ws_, int64_t cap)
        : e(e_), s(s_), arena(cap), dry(dry_), ws(ws_), main_s(s_), sw(Switches::read()) {
        // DFFW_DEBUG_FLAGS bit 0 ("no footprint fill") is withdrawn: kernels then contract uninitialised LDS and the run ended in an abort of the
        // whole process (profiles/r04_ablation_conv_tile_phases.txt) -- an error code instead; bits 1 (no MFMA loop) and 2 (no stores) remain
        if (sw.debug_flags & 1) err = fail(DFFW_EINVAL, "DFFW_DEBUG_FLAGS bit 0 (skip the footprint fill) is not supported; use 2 (no MFMA loop), 4 (no stores) or 6");
    }

    bool ok() const { return err == DFFW_OK; }

    void enable_concurrency() {
        concurrent = true;
        if (!dry && ok() && e->ensure_side() != DFFW_OK) err = DFFW_EHIP;
    }
    void fork(int k) {   // side stream k continues from the current point of the main stream
        if (!concurrent || dry || !ok()) return;
        hipEvent_t v = e->ev[ev_next++ % dffw_engine::NEV];
        check(hipEventRecord(v, main_s), "fork record");
        check(hipStreamWaitEvent(e->side[k], v, 0), "fork wait");
        fork_mark(k);
    }
    void on(int k) { s = (concurrent && !dry && k >= 0) ? e->side[k] : main_s; }   // stream of the following launches
    void fork_mark(int k) { side_open |= 1u << k; }
    void join(int k) {   // the main stream waits for everything queued on side stream k
        if (!concurrent || dry || !e->side[k]) return;
        // (also after an error: kernels already queued on the side stream use the workspace, and the caller is free to
        // recycle it as soon as the main stream is done)
        hipEvent_t v = e->ev[ev_next++ % dffw_engine::NEV];
        const hipError_t h1 = hipEventRecord(v, e->side[k]);
        const hipError_t h2 = h1 == hipSuccess ? hipStreamWaitEvent(main_s, v, 0) : h1;
        side_open &= ~(1u << k);
        if (ok()) check(h2, "join");
    }
    ~Run() {   // a forked section left through an error path: join whatever is still open
        for (int k = 0; k < dffw_engine::NSIDE; ++k)
            if (side_open & (1u << k)) join(k);
    }
    void release_deferred() {
        for (void *p : deferred) arena.release(dry ? (int64_t)(uintptr_t)p - 256 : (int64_t)((char *)p - ws));
        deferred.clear();
    }

    void *raw(int64_t bytes) {
        if (!ok()) return nullptr;
        const int64_t off = arena.alloc(bytes);
        if (off < 0) {
            err = fail(DFFW_ENOMEM, "workspace too small (need more than the %lld bytes given)", (long long)bytes);
            return nullptr;
        }
        return dry ? (void *)(uintptr_t)(off + 256) : (void *)(ws + off);
    }
    void drop_raw(void *p) {
        if (!p) return;
        if (forked) {
            deferred.push_back(p);
            return;
        }
        arena.release(dry ? (int64_t)(uintptr_t)p - 256 : (int64_t)((char *)p - ws));
    }
    Act act(int B, int N, int H, int W, int C) {
        Act a;
        a.B = B; a.N = N; a.H = H; a.W = W; a.C = C;
        a.p = (uint16_t *)raw(a.pixels() * prec_parts(e->prec) * C * (int64_t)sizeof(uint16_t));
        return a;
    }
    void drop(Act &a) {
        drop_raw(a.p);
        a.p = nullptr;
    }
    void check(hipError_t h, const char *what) {
        if (ok() && h != hipSuccess) err = fail(DFFW_EHIP, "%s: %s", what, hipGetErrorString(h));
    }
    // bracket one launch with HIP events on the launch stream (profiling mode only)
    void prof_begin(const char *kernel, const std::string &layer, double flops, double bytes) {
        if (!e->profiling || dry || !ok()) return;
        ProfRec pr;
        pr.kernel = kernel;
        pr.layer = layer;
        pr.flops = flops;
        pr.bytes = bytes;
        check(hipEventCreate(&pr.e0), "hipEventCreate");
        check(hipEventCreate(&pr.e1), "hipEventCreate");
        if (ok()) check(hipEventRecord(pr.e0, s), "hipEventRecord");
        e->recs.push_back(pr);
    }
    void prof_end() {
        if (!e->profiling || dry || !ok() || e->recs.empty()) return;
        check(hipEventRecord(e->recs.back().e1, s), "hipEventRecord");
    }
    double elem_bytes() const { return 2.0 * prec_parts(e->prec); }

    // step timeline of a persistent streaming kernel (library built with `make TRACE=1`; DFFW_TRACE_LAYER=<layer> DFFW_TRACE_OUT=<file>):
    // [workgroup][wave][32 steps][8] u64 of s_memtime stamps (StepTrace, dffw_device.h; tools/trace_steps.py)
    unsigned long long *trace_dev = nullptr;
    size_t trace_n = 0;
    unsigned long long *trace_begin(const std::string &layer, int wgs, int waves) {
        if (dry || !ok() || !sw.trace_layer || !sw.trace_out || layer != sw.trace_layer) return nullptr;
        trace_n = (size_t)wgs * waves * 32 * 8;
        check(hipMalloc((void **)&trace_dev, trace_n * 8), "trace alloc");
        if (ok()) check(hipMemsetAsync(trace_dev, 0, trace_n * 8, s), "trace memset");
        return ok() ? trace_dev : nullptr;
    }
    void trace_end() {
        if (!trace_dev) return;
        if (ok()) {
            std::vector<unsigned long long> host(trace_n);
            check(hipStreamSynchronize(s), "trace sync");
            check(hipMemcpy(host.data(), trace_dev, trace_n * 8, hipMemcpyDeviceToHost), "trace copy");
            if (FILE *f = fopen(sw.trace_out, "wb")) {
                fwrite(host.data(), 8, host.size(), f);
                fclose(f);
            }
        }
        (void)hipFree(trace_dev);
        trace_dev = nullptr;
    }

    // does the LDS-tiled kernel serve layer `name` for an output grid of gH x gW (same test as in conv())?
    bool tiled(const std::string &name, int gH, int gW) const {
        auto it = e->convs.find(name);
        if (it == e->convs.end()) return false;
        const TileCfg *c = it->second.tile.cfg;
        return c && !sw.on(SW_NO_TILE) && gW * 2 >= c->tx && gH * 2 >= c->ty;
    }

    // conv_tile's row-sums variant serves per-slice layer `name` on a (B,N,H,W) volume: tiled, an instantiation exists, whole row
    // segments of 16 pixels, and enough tiles that conv() will not split the output channels over the grid
    bool sums_conv_ok(const std::string &name, int B, int N, int H, int W) const {
        if (!tiled(name, H, W)) return false;
        const TileCfg *c = e->convs.find(name)->second.tile.cfg;
        if (!tile_cfg_has_sums(c) || W % c->tx) return false;
        const int64_t tiles = (int64_t)B * ((N + c->tz - 1) / c->tz) * ((H + c->ty - 1) / c->ty) * (W / c->tx);
        return tiles >= 256;
    }

    Act conv(const std::string &name, const Act &in0, const ConvOpt &o = ConvOpt()) {
        Act out;
        if (!ok()) return out;
        auto it = e->convs.find(name);
        if (it == e->convs.end()) {
            err = fail(DFFW_EINVAL, "no packed layer %s", name.c_str());
            return out;
        }
        const PackedConv &pc = it->second;
        const LayerDef &L = pc.def;
        const int cin = in0.C + (o.in1 ? o.in1->C : 0);
        const int cin_pad = pc.cin_all;
        if (cin != cin_pad) {
            err = fail(DFFW_EINVAL, "layer %s expects %d input channels, got %d", name.c_str(), cin_pad, cin);
            return out;
        }
        int Ho, Wo;
        if (L.transposed) {
            Ho = in0.H * 2;
            Wo = in0.W * 2;
        } else {
            const int win = (L.kh == 9 && L.cin == 3) ? in0.W - 2 : in0.W;  // stem input is the (W+2)-wide paired volume
            Ho = (in0.H + 2 * L.ph - L.dh * (L.kh - 1) - 1) / L.sh + 1;
            Wo = (win + 2 * L.pw - L.dw * (L.kw - 1) - 1) / L.sw + 1;
        }
        const int No = in0.N + 2 * L.pd - (L.kd - 1);
        if (o.outf == nullptr && !o.discard && !o.sums) out = act(in0.B, No, Ho, Wo, L.cout);
        else { out.B = in0.B; out.N = No; out.H = Ho; out.W = Wo; out.C = L.cout; }
        const float *cls_w = nullptr;
        if (o.cls) {
            auto ic = e->convs.find(o.cls);
            if (ic == e->convs.end() || !ic->second.w32 || ic->second.def.cin != L.cout || ic->second.def.cout != 1) {
                err = fail(DFFW_EINVAL, "cannot fuse classifier %s into %s", o.cls, name.c_str());
                return out;
            }
            cls_w = ic->second.w32;
        }
        if (o.out_pre) *o.out_pre = act(in0.B, No, Ho, Wo, L.cout);
        if (!ok()) return out;
        // (the dry run that sizes the workspace continues through the launch planning below: split-K adds a scratch block)

        ConvArgs a;
        memset(&a, 0, sizeof a);
        a.in0 = in0.p;
        a.C0 = in0.C;
        a.in1 = o.in1 ? o.in1->p : in0.p;
        a.C1 = o.in1 ? o.in1->C : 0;
        a.B = in0.B; a.Ni = in0.N; a.Hi = in0.H; a.Wi = in0.W;
        a.No = No; a.Ho = Ho; a.Wo = Wo;
        a.Cout = L.cout;
        a.bias = pc.bias;
        a.res0 = o.res0 ? o.res0->p : nullptr;
        a.res1 = o.res1 ? o.res1->p : nullptr;
        a.res_bcast = o.res_bcast ? 1 : 0;
        if (o.res_bcast && !(L.kd == 1 && L.kh == 3 && !L.transposed && L.sh == 1 && L.cout >= 16)) {
            err = fail(DFFW_EINVAL, "slice-broadcast residual is only implemented for the per-slice 1x3x3 convs (layer %s)", name.c_str());
            return out;
        }
        a.out = out.p;
        a.out_pre = o.out_pre ? o.out_pre->p : nullptr;
        a.outf = o.outf;
        a.fs32 = o.fs32;
        a.outf_ch = o.outf_ch;
        a.outf_plane = (int64_t)No * Ho * Wo;
        a.cls_w = cls_w;
        a.cls_out = o.cls_out;
        a.relu = o.relu;
        if (!dry && e->ensure_zero_page() != DFFW_OK) { err = DFFW_EHIP; return out; }
        a.zero = e->zero_page;
        a.dbg = (sw.debug_flags & 6) | sw.path_bits();   // ablation switches (2 no MFMA loop, 4 no stores) + the launchers' path switches
        if (o.raw) a.dbg |= DFFW_ARGS_RAW;   // fs32 then points to the RawStack descriptor in device memory
        if (o.sums) {
            if (!sums_conv_ok(name, in0.B, in0.N, in0.H, in0.W) || o.relu != 1 || o.res0 || o.res1 || o.cls || o.out_pre || o.in1) {
                err = fail(DFFW_EINVAL, "layer %s has no row-sums kernel for this shape / epilogue", name.c_str());
                return out;
            }
            a.outf = o.sums;
            a.dbg |= DFFW_ARGS_SUMS;
        }
        // transposed 32 / 64 -> 32 / 64 (deconv_1, dres2.conv5 / conv6, dres3.conv5, SPP conv9) on 8 x 8 columns of the input grid: the streaming kernel with the
        // filter split over the waves by output phase; a unit = (column, 32-channel output half)
        if (pc.wrollt && !o.in1 && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_ROLLT)) {
            int tty, ttx;
            rollt_tile(L.cout, &tty, &ttx);
            const int cols = ((in0.H + tty - 1) / tty) * ((in0.W + ttx - 1) / ttx);   // (partial columns are predicated in the kernel)
            ConvArgs ak = a;
            ak.Ng = in0.N; ak.Hg = in0.H; ak.Wg = in0.W;
            ak.M = (int64_t)ak.B * in0.N * in0.H * in0.W;
            // (column, output half) units from which the kernel beats conv_tile / conv_roll_t32, measured at batch 8 / 16 / 32 (profiles/r06_rollt_thresholds.txt): 128 for the
            // 8-wave forms -- half the CUs busy, and still 0.046 vs 0.058 ms on deconv_1 at batch 8, 0.052 vs 0.072 on SPP conv9 at batch 32; 64 units lose --, 192 for the 4-wave form
            const int64_t rt_units = (int64_t)in0.B * cols * std::max(1, L.cout / 32);
            const bool rt_four = cin_pad == 32 && L.cout != 16;
            if (rt_units >= (int64_t)sw.rollt_min_units * (rt_four ? 3 : 2) / 2 && rollt_ok(e->prec, ak)) {
                if (dry) return out;
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wrollt;
                t.tiles_y = (in0.H + tty - 1) / tty;
                t.tiles_x = (in0.W + ttx - 1) / ttx;
                t.zsplit = 1;
                t.total_tiles = in0.B * cols;
                t.wgs = sw.roll_wgs;
                char kn[96];
                conv_rollt_kernel_name(ak, kn, sizeof kn);
                g_last_kernel = kn;
                const double opx = (double)out.B * No * Ho * Wo;
                // the fused classifier's two partial dots per pixel are ADDED to the score volume
                if (ak.cls_w) check(hipMemsetAsync(ak.cls_out, 0, (size_t)(opx * 4.0), s), "score memset");
                if (e->profiling) {
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes()
                                         + opx * L.cout * elem_bytes() * ((o.discard ? 0 : 1) + (o.out_pre ? 1 : 0) + (o.res0 ? 1 : 0)) + (o.cls ? opx * 4.0 : 0.0)
                                         + 27.0 * L.cin * L.cout * elem_bytes();
                    prof_begin(kn, name, 2.0 * (double)ak.M * 27.0 * L.cin * L.cout, bytes);
                }
                check(launch_conv_rollt(ak, t, s), name.c_str());
                prof_end();
                return out;
            }
        }
        // transposed 32 -> 16 (deconv_2, dres3.conv6): two sweeps of conv_roll_t32, one per output row phase
        if (pc.wroll_t32 && (in0.C == 32 || in0.C == 16) && !o.in1 && !o.res_bcast && !o.res1 && !o.outf && in0.H % 8 == 0 && in0.W % 16 == 0 &&
            (int64_t)in0.B * (in0.H / 8) * (in0.W / 16) >= sw.roll_min_units && !sw.on(SW_NO_ROLL)) {
            if (dry) return out;
            a.Ng = in0.N; a.Hg = in0.H; a.Wg = in0.W;
            a.M = (int64_t)a.B * in0.N * in0.H * in0.W;
            a.dbg &= (6 | DFFW_ARGS_NO_LEAN_TILE | DFFW_ARGS_NO_LEAN_ROLL | DFFW_ARGS_NO_ROLLX | DFFW_ARGS_NO_ROLLK | DFFW_ARGS_NO_SLICE32);   // (the ablation bits these kernels know + the launchers' path switches)
            for (int py = 0; py < 2; ++py) {
                int rty, rtx;
                roll_t32_tile(py, &rty, &rtx);
                RollArgs t;
                memset(&t, 0, sizeof t);
                const int parts = prec_parts(e->prec);
                t.wroll = pc.wroll_t32 + (size_t)(py ? ROLL_CHUNKS_T32_0 : 0) * parts * 512;
                t.tiles_y = in0.H / rty;
                t.tiles_x = in0.W / rtx;
                t.zsplit = 1;
                t.total_tiles = in0.B * t.tiles_y * t.tiles_x;
                t.wgs = sw.roll_wgs;
                char kn[96];
                conv_roll_t32_kernel_name(e->prec, py, a, kn, sizeof kn);
                g_last_kernel = kn;
                if (e->profiling) {
                    const double opx = (double)out.B * No * Ho * Wo * 0.5;   // this sweep's output pixels
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes()
                                         + opx * L.cout * elem_bytes() * ((o.discard ? 0 : 1) + (o.out_pre ? 1 : 0) + (o.res0 ? 1 : 0)) + (o.cls ? opx * 4.0 : 0.0);
                    prof_begin(kn, name + (py ? " (odd rows)" : " (even rows)"), 2.0 * (double)a.M * (py ? 18.0 : 9.0) * L.cin * L.cout, bytes);
                }
                check(launch_conv_roll_t32(e->prec, py, a, t, s), name.c_str());
                prof_end();
            }
            return out;
        }
        // strided 3x3x3 over 16 / 32 channels (FM_conv2.0.stride_conv, dres3.conv1, dres4.conv3; dres3.conv3, dres2.conv1, SPP conv1):
        // rolling window with whole pixel records
        if (pc.wroll_s2 && !L.transposed && L.sh == 2 && (in0.C == 16 || in0.C == 32) && !o.in1 && !o.res1 && !o.res_bcast && !o.outf && !o.out_pre &&
            !o.cls && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_ROLL_S2) &&
            (L.cout <= 32 || !sw.on(SW_NO_ROLL_S2_WIDE))) {   // 32 -> 64 as two launches: level with conv_tile in r02, 4-7 % faster since the r04 row-pitch fix of conv_roll_s2
            const int khn = in0.C / 16;
            const int ntk = (khn == 2 || L.cout >= 32) ? 2 : 1;     // output tiles per launch
            const int nlaunch = (L.cout / 16 + ntk - 1) / ntk;
            int sty, stx;
            s2_roll_tile(ntk, &sty, &stx);
            if ((L.cout / 16) % ntk == 0 && Ho % sty == 0 && Wo % stx == 0 && in0.H == 2 * Ho && in0.W == 2 * Wo &&
                (int64_t)in0.B * (Ho / sty) * (Wo / stx) >= sw.roll_min_units) {
                if (dry) return out;
                a.Ng = No; a.Hg = Ho; a.Wg = Wo;
                a.M = (int64_t)a.B * No * Ho * Wo;
                a.dbg &= (6 | DFFW_ARGS_NO_LEAN_TILE | DFFW_ARGS_NO_LEAN_ROLL | DFFW_ARGS_NO_ROLLX | DFFW_ARGS_NO_ROLLK | DFFW_ARGS_NO_SLICE32);   // (the ablation bits these kernels know + the launchers' path switches)
                for (int li = 0; li < nlaunch; ++li) {
                    RollArgs t;
                    memset(&t, 0, sizeof t);
                    t.wroll = pc.wroll_s2;
                    t.tiles_y = Ho / sty;
                    t.tiles_x = Wo / stx;
                    t.zsplit = 1;
                    t.total_tiles = in0.B * t.tiles_y * t.tiles_x;
                    t.wgs = sw.roll_wgs;
                    t.pair = li * ntk;     // first 16-channel output tile of this launch
                    char kn[96];
                    conv_roll_s2_kernel_name(e->prec, ntk, khn, a, kn, sizeof kn);
                    g_last_kernel = kn;
                    if (e->profiling) {
                        const double opx = (double)out.B * No * Ho * Wo;
                        prof_begin(kn, nlaunch > 1 ? name + (li ? " (upper output channels)" : " (lower output channels)") : name,
                                   2.0 * opx * 27.0 * L.cin * L.cout / nlaunch,
                                   ((double)in0.pixels() * L.cin + opx * L.cout / nlaunch * (1 + (o.res0 ? 1 : 0))) * elem_bytes() + 27.0 * L.cin * L.cout / nlaunch * elem_bytes());
                    }
                    check(launch_conv_roll_s2(e->prec, ntk, khn, a, t, s), name.c_str());
                    prof_end();
                }
                return out;
            }
        }
        // strided 3x3x3 8 -> 16 (dres4.conv1): the single-branch form of conv_roll_efd
        {
            int ety, etx;
            efd_roll_tile(&ety, &etx);
            if (pc.wroll8 && !L.transposed && L.sh == 2 && in0.C == 8 && !o.in1 && !o.res0 && !o.res1 && !o.res_bcast && !o.outf && !o.out_pre &&
                !o.cls && Ho % ety == 0 && Wo % etx == 0 && (int64_t)in0.B * (Ho / ety) * (Wo / etx) >= sw.roll_min_units && !sw.on(SW_NO_ROLL) &&
                !sw.on(SW_NO_ROLL_S2)) {
                if (dry) return out;
                a.Ng = No; a.Hg = Ho; a.Wg = Wo;
                a.M = (int64_t)a.B * No * Ho * Wo;
                a.dbg &= (6 | DFFW_ARGS_NO_LEAN_TILE | DFFW_ARGS_NO_LEAN_ROLL | DFFW_ARGS_NO_ROLLX | DFFW_ARGS_NO_ROLLK | DFFW_ARGS_NO_SLICE32);   // (the ablation bits these kernels know + the launchers' path switches)
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wroll8;
                t.tiles_y = Ho / ety;
                t.tiles_x = Wo / etx;
                t.zsplit = 1;
                t.total_tiles = in0.B * t.tiles_y * t.tiles_x;
                t.wgs = sw.roll_wgs;
                char kn[96];
                conv_roll_efd_kernel_name(e->prec, a, false, kn, sizeof kn);
                g_last_kernel = kn;
                if (e->profiling) {
                    const double opx = (double)out.B * No * Ho * Wo;
                    prof_begin(kn, name, 2.0 * opx * 27.0 * L.cin * L.cout,
                               ((double)in0.pixels() * L.cin + opx * L.cout) * elem_bytes() + 27.0 * L.cin * L.cout * elem_bytes());
                }
                check(launch_conv_roll_efd(e->prec, a, t, s), name.c_str());
                prof_end();
                return out;
            }
        }
        // ... and its transposed sibling (16 -> 8 channels), tiled over the input grid
        {
            int rty, rtx;
            roll_tile(&rty, &rtx);
            const int cols = (in0.H / rty) * (in0.W / rtx);
            if (pc.wroll_t && in0.H % rty == 0 && in0.W % rtx == 0 && in0.C == 16 && !o.in1 && !o.res_bcast && !o.res1 && !o.outf &&
                (int64_t)in0.B * cols >= sw.roll_min_units && !sw.on(SW_NO_ROLL)) {
                if (dry) return out;
                a.Ng = in0.N; a.Hg = in0.H; a.Wg = in0.W;
                a.M = (int64_t)a.B * in0.N * in0.H * in0.W;
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wroll_t;
                t.tiles_y = in0.H / rty;
                t.tiles_x = in0.W / rtx;
                t.zsplit = ((int64_t)in0.B * cols < 1024 && No >= 8) ? 2 : 1;
                if (sw.roll_zsplit >= 1 && sw.roll_zsplit <= No) t.zsplit = sw.roll_zsplit;
                t.total_tiles = in0.B * t.zsplit * cols;
                t.wgs = 0;
                if (sw.roll_wgs) t.wgs = sw.roll_wgs;
                t.pair = 1;
                char kn[96];
                conv_roll_t_kernel_name(e->prec, a, kn, sizeof kn);
                g_last_kernel = kn;
                if (e->profiling) {
                    const double opx = (double)out.B * No * Ho * Wo;
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes()
                                         + opx * L.cout * elem_bytes() * ((o.discard ? 0 : 1) + (o.out_pre ? 1 : 0))
                                         + opx * L.cout * elem_bytes() * (o.res0 ? 1 : 0) + (o.cls ? opx * 4.0 : 0.0)
                                         + 27.0 * L.cin * L.cout * elem_bytes();
                    prof_begin(kn, name, 2.0 * (double)a.M * 27.0 * L.cin * L.cout, bytes);
                }
                check(launch_conv_roll_t(e->prec, a, t, s), name.c_str());
                prof_end();
                return out;
            }
        }
        // the stem straight from the focal stack: pixel-pair form (half the MFMAs and LDS reads of the per-pixel kernel)
        const bool stem_pair = o.fs32 && pc.tile_pair.cfg && pc.bias_pair && Wo % pc.tile_pair.cfg->tx == 0 && Ho % pc.tile_pair.cfg->ty == 0 &&
                               !sw.on(SW_NO_STEM_PAIR);
        if (stem_pair) a.bias = pc.bias_pair;
        const int gW = L.transposed ? in0.W : Wo, gH = L.transposed ? in0.H : Ho;
        // the 5 x 8 x 8 block (its packs of layers with more than 4 output tiles split the output channels over grid.y: not with a fused classifier,
        // whose partial dot spans all of a pixel's channels, nor under DFFW_NO_SPLIT), with enough samples / tiles to fill the chip:
        // (a) grids at most 8 x 8 (the 1/32-resolution pyramid layers at 256 x 256, round 4);  (b) round 5: stride-1 layers with 128 output channels on grids up to
        // DFFW_NARROW_MAX (40) wide -- End_to_End's 30 x 40 pyramid level at batch 8: `combine2` / `conv4` ran on the 4 x 4 x 8 block with all 128 output
        // channels per workgroup, re-streaming the filter for 128 grid points at a time (0.24 -> 0.13 ms, `conv4` 0.16 -> 0.09; the 64-output layers of those levels gain 2-8 % on it at that shape and lose as much at others: left alone)
        const bool narrow_splits = pc.tile_narrow.cfg && pc.nt > pc.tile_narrow.cfg->nt;
        const int gN = L.transposed ? in0.N : No;
        bool narrow = !stem_pair && pc.tile_narrow.cfg && !sw.on(SW_NO_NARROW) && !(narrow_splits && (o.cls || sw.on(SW_NO_SPLIT))) &&
                      (int64_t)in0.B * ((gN + 4) / 5) * ((gH + 7) / 8) * ((gW + 7) / 8) * pc.nt >= 256;
        if (narrow && !(gW <= 8 && gH <= 8)) narrow = !L.transposed && pc.nt >= 8 && gW <= sw.narrow_max && gH <= sw.narrow_max;
        const TilePack &tp = stem_pair ? pc.tile_pair : (narrow ? pc.tile_narrow : pc.tile);
        // 32 -> 16 channels on whole 8 x 16 columns: the pipelined rolling window with the contraction split over the two input halves
        {
            const int cols = (Ho / 8) * (Wo / 16);
            const bool halves = o.in1 ? (in0.C == 16 && o.in1->C == 16) : in0.C == 32;
            if (pc.wroll_k2 && halves && Ho % 8 == 0 && Wo % 16 == 0 && (int64_t)in0.B * cols >= sw.roll_min_units && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_ROLLX)) {
                ConvArgs ak = a;
                ak.Ng = No; ak.Hg = Ho; ak.Wg = Wo;
                ak.M = (int64_t)ak.B * No * Ho * Wo;
                if (rollx_k2_ok(e->prec, ak)) {
                    if (dry) return out;
                    RollArgs t;
                    memset(&t, 0, sizeof t);
                    t.wroll = pc.wroll_k2;
                    t.tiles_y = Ho / 8;
                    t.tiles_x = Wo / 16;
                    t.zsplit = ((int64_t)in0.B * cols < 512 && No >= 8) ? 2 : 1;
                    if (sw.roll_zsplit >= 1 && sw.roll_zsplit <= No) t.zsplit = sw.roll_zsplit;
                    t.total_tiles = in0.B * t.zsplit * cols;
                    t.wgs = sw.roll_wgs;
                    char kn[96];
                    conv_rollx_k2_kernel_name(ak, kn, sizeof kn);
                    g_last_kernel = kn;
                    if (e->profiling) {
                        const double opx = (double)out.B * No * Ho * Wo;
                        const double bytes = (double)in0.pixels() * L.cin * elem_bytes() + opx * L.cout * elem_bytes() + 27.0 * L.cin * L.cout * elem_bytes();
                        prof_begin(kn, name, 2.0 * opx * 27.0 * L.cin * L.cout, bytes);
                    }
                    check(launch_conv_rollx_k2(ak, t, s), name.c_str());
                    prof_end();
                    return out;
                }
            }
        }
        // per-slice 1x3x3, 32 -> 32 channels on whole 8 x 16 columns: the streaming kernel with the filter resident in every wave
        if (pc.wslice32 && !o.in1 && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_SLICE32)) {
            int sty, stx;
            slice32_tile(&sty, &stx);
            const int cols = (Ho / sty) * (Wo / stx);
            ConvArgs ak = a;
            ak.Ng = No; ak.Hg = Ho; ak.Wg = Wo;
            ak.M = (int64_t)ak.B * No * Ho * Wo;
            if (Ho % sty == 0 && Wo % stx == 0 && (int64_t)in0.B * cols >= sw.roll_min_units && slice32_ok(e->prec, ak)) {
                if (dry) return out;
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wslice32;
                t.tiles_y = Ho / sty;
                t.tiles_x = Wo / stx;
                t.zsplit = 1;
                t.total_tiles = in0.B * cols;
                t.wgs = sw.roll_wgs;
                char kn[96];
                conv_slice32_kernel_name(ak, kn, sizeof kn);
                g_last_kernel = kn;
                if (e->profiling) {
                    const double opx = (double)out.B * No * Ho * Wo;
                    const double obytes = o.sums ? opx / 16.0 * 3.0 * L.cout * 4.0 : opx * L.cout * elem_bytes() * (1 + (o.res0 ? 1 : 0));
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes() + obytes + 9.0 * L.cin * L.cout * elem_bytes();
                    prof_begin(kn, name, 2.0 * opx * 9.0 * L.cin * L.cout, bytes);
                }
                check(launch_conv_slice32(ak, t, s), name.c_str());
                prof_end();
                return out;
            }
        }
        // per-slice 1x3x3, 64 -> 64 channels on whole 8 x 16 columns: the streaming kernel with one output tile's filter resident per wave
        if (pc.wslice64 && (pc.slice_cat ? (o.in1 && o.in1->C == 32) : !o.in1) && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_SLICE32)) {
            int sty, stx;
            slice32_tile(&sty, &stx);
            const int cols = (Ho / sty) * (Wo / stx);
            ConvArgs ak = a;
            ak.Ng = No; ak.Hg = Ho; ak.Wg = Wo;
            ak.M = (int64_t)ak.B * No * Ho * Wo;
            if (Ho % sty == 0 && Wo % stx == 0 && (int64_t)in0.B * cols >= sw.roll_min_units && slice64_ok(e->prec, ak)) {
                if (dry) return out;
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wslice64;
                t.tiles_y = Ho / sty;
                t.tiles_x = Wo / stx;
                t.zsplit = 1;
                t.total_tiles = in0.B * cols;
                t.wgs = sw.roll_wgs;
                char kn[96];
                conv_slice64_kernel_name(ak, kn, sizeof kn);
                g_last_kernel = kn;
                if (e->profiling) {
                    const double opx = (double)out.B * No * Ho * Wo;
                    const double obytes = o.sums ? opx / 16.0 * 3.0 * L.cout * 4.0 : opx * L.cout * elem_bytes();
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes() + obytes + 9.0 * L.cin * L.cout * elem_bytes();
                    prof_begin(kn, name, 2.0 * opx * 9.0 * L.cin * L.cout, bytes);
                }
                check(launch_conv_slice64(ak, t, s), name.c_str());
                prof_end();
                return out;
            }
        }
        // 32 / 64 -> 32 / 64 channels on whole 8 x 8 columns: the K-split rolling window (one launch per 32 output channels)
        if (pc.wrollk && !sw.on(SW_NO_ROLL) && !sw.on(SW_NO_ROLLK)) {
            int kty, ktx;
            rollk_tile(&kty, &ktx);
            const int cols = ((Ho + kty - 1) / kty) * ((Wo + ktx - 1) / ktx);   // (partial columns at the bottom / right edge are predicated in the kernel)
            ConvArgs ak = a;
            ak.Ng = No; ak.Hg = Ho; ak.Wg = Wo;
            ak.M = (int64_t)ak.B * No * Ho * Wo;
            // 64 output channels = two 32-channel halves as grid.y of ONE launch (round 6: the 16 x 16-grid layers `dres16_*`, `conv2`, `dres2.conv4` at batch 32
            // are 128 columns x 2 halves = one unit per CU and now take this kernel; DFFW_ROLLK_MERGE_BELOW=1: two launches)
            const int npair = L.cout / 32;
            const bool merged = npair == 2 && (int64_t)in0.B * cols < sw.rollk_merge_below;
            if ((int64_t)in0.B * cols * (merged ? npair : 1) >= sw.roll_min_units && rollk_waves(e->prec, ak) == cin_pad / 8) {
                if (dry) return out;
                const int nlaunch = merged ? 1 : npair;
                for (int op = 0; op < nlaunch; ++op) {
                    RollArgs t;
                    memset(&t, 0, sizeof t);
                    t.wroll = pc.wrollk + (size_t)op * (cin_pad / 8) * ROLLK_CHUNKS * 2 * prec_parts(e->prec) * 512;
                    t.tiles_y = (Ho + kty - 1) / kty;
                    t.tiles_x = (Wo + ktx - 1) / ktx;
                    // a sample's slices as two ranges where whole columns leave the chip short of workgroups (16 waves per CU: 256 8-wave / 512 4-wave units)
                    t.zsplit = ((int64_t)in0.B * cols * (merged ? npair : 1) < (cin_pad == 64 && merged ? 256 : 512) && No >= 8) ? 2 : 1;
                    if (sw.roll_zsplit >= 1 && sw.roll_zsplit <= No) t.zsplit = sw.roll_zsplit;
                    t.total_tiles = in0.B * t.zsplit * cols;
                    t.wgs = sw.roll_wgs;
                    t.pair = merged ? -1 : op * 2;     // first 16-channel output tile of this launch (-1: every half, as grid.y)
                    char kn[96];
                    conv_rollk_kernel_name(ak, kn, sizeof kn);
                    g_last_kernel = kn;
                    if (e->profiling) {
                        const double opx = (double)out.B * No * Ho * Wo;
                        const double bytes = (double)in0.pixels() * L.cin * elem_bytes() + opx * L.cout / nlaunch * elem_bytes() * (1 + (o.res0 ? 1 : 0)) +
                                             27.0 * L.cin * L.cout / nlaunch * elem_bytes();
                        prof_begin(kn, nlaunch > 1 ? name + (op ? " (upper output channels)" : " (lower output channels)") : name,
                                   2.0 * opx * 27.0 * L.cin * L.cout / nlaunch, bytes);
                    }
                    check(launch_conv_rollk(ak, t, s), name.c_str());
                    prof_end();
                }
                return out;
            }
        }
        // rolling-window kernel: 16-channel 3x3x3 stride-1 layers whose grid is whole columns and fills the chip
        {
            int rty, rtx;
            roll_tile(&rty, &rtx);
            const int cols = (Ho / rty) * (Wo / rtx);
            if (pc.wroll && Ho % rty == 0 && Wo % rtx == 0 && in0.C % 8 == 0 && (!o.in1 || o.in1->C == in0.C) && !o.res_bcast && !o.res1 &&
                (int64_t)in0.B * cols >= sw.roll_min_units && !sw.on(SW_NO_ROLL)) {
                if (dry) return out;
                a.Ng = No; a.Hg = Ho; a.Wg = Wo;
                a.M = (int64_t)a.B * No * Ho * Wo;
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = pc.wroll;
                t.tiles_y = Ho / rty;
                t.tiles_x = Wo / rtx;
                t.zsplit = ((int64_t)in0.B * cols < 1024 && No >= 8) ? 2 : 1;
                if (sw.roll_zsplit >= 1 && sw.roll_zsplit <= No) t.zsplit = sw.roll_zsplit;
                t.total_tiles = in0.B * t.zsplit * cols;
                t.wgs = 0;
                t.pair = pc.roll_pair ? 1 : 0;
                if (sw.roll_wgs) t.wgs = sw.roll_wgs;
                {
                    char kn[96];
                    conv_roll_kernel_name(e->prec, a, pc.roll_pair, kn, sizeof kn);
                    g_last_kernel = kn;
                }
                if (e->profiling) {
                    char kn[96];
                    conv_roll_kernel_name(e->prec, a, pc.roll_pair, kn, sizeof kn);
                    const double opx = (double)out.B * No * Ho * Wo;
                    const double bytes = (double)in0.pixels() * L.cin * elem_bytes()
                                         + opx * L.cout * (o.outf ? 4.0 : elem_bytes() * (o.out_pre ? 2 : 1))
                                         + opx * L.cout * elem_bytes() * ((o.res0 ? 1 : 0) + (o.res1 ? 1 : 0))
                                         + 27.0 * L.cin * L.cout * elem_bytes();
                    prof_begin(kn, name, 2.0 * opx * 27.0 * L.cin * L.cout, bytes);
                }
                a.trace = trace_begin(name, 1024, 4);
                check(launch_conv_roll(e->prec, a, t, s), name.c_str());
                prof_end();
                trace_end();
                return out;
            }
        }
        // small grids (the low-resolution pyramid at batch 1): conv_small's one-workgroup-per-(16 points, 16 channels) split of the
        // whole layer beats an LDS tile that a few workgroups walk stage by stage (+ a split-K finish launch)
        const int64_t small_units = ((int64_t)in0.B * (L.transposed ? in0.N : No) * gH * gW + 15) / 16 * pc.nt;
        const bool prefer_small = small_units <= sw.small_max_units && !o.cls && !o.fs32 && !sw.on(SW_NO_SMALL) && !stem_pair;
        const bool use_tile = tp.cfg && !sw.on(SW_NO_TILE) && gW * 2 >= tp.cfg->tx && gH * 2 >= tp.cfg->ty &&
                              in0.C % 8 == 0 && (!o.in1 || o.in1->C % 8 == 0) && !prefer_small;
        if (use_tile) {
            const TileCfg *cfg = tp.cfg;   // may be replaced by a narrower instantiation of the same tile (channel split)
            a.Ng = L.transposed ? in0.N : No;
            a.Hg = gH;
            a.Wg = gW;
            a.sy = a.sx = L.transposed ? 1 : L.sh;
            a.osy = a.osx = L.transposed ? 2 : 1;
            a.M = (int64_t)a.B * a.Ng * a.Hg * a.Wg;
            TileArgs t;
            memset(&t, 0, sizeof t);
            t.npass = tp.npass;
            t.nstage = tp.nstage;
            double flops = 0;
            for (int ps = 0; ps < tp.npass; ++ps) {
                t.KC[ps] = tp.KC[ps];
                t.tab[ps] = tp.tab[ps];
                t.wpk[ps] = tp.wpk[ps];
                t.ooy[ps] = tp.ooy[ps];
                t.oox[ps] = tp.oox[ps];
                flops += 2.0 * (double)a.M * (L.transposed ? tp.ntaps[ps] : L.kd * L.kh * L.kw) * L.cin * L.cout;
            }
            t.tiles_z = (a.Ng + cfg->tz - 1) / cfg->tz;
            t.tiles_y = (a.Hg + cfg->ty - 1) / cfg->ty;
            t.tiles_x = (a.Wg + cfg->tx - 1) / cfg->tx;
            t.total_tiles = a.B * t.tiles_z * t.tiles_y * t.tiles_x;
            t.nt_total = pc.nt;
            t.nsplit = pc.nt / cfg->nt;   // (1, except the narrow packs of layers with more than 4 output tiles)
            // few-tile layers (the 1/16..1/32-resolution pyramid, or batch 1): split the output channels over
            // grid.y so that at least ~one workgroup per CU exists
            // (3x3x3 stride-1 and transposed layers also at exactly one tile per CU -- the 16x16-grid layers at batch 32: two 32-channel
            // workgroups per tile keep three workgroups resident instead of two, -10 % on those layers; the stride-2 layers lose 40 % with it)
            // (transposed layers with 64 outputs: the 4-output-tile block runs at 127 TFLOP/s where two launches' worth of 2-tile workgroups run at 212 -- measured on End_to_End's
            // `dres2.conv5`, 384 tiles at batch 8 --, so they split up to 1024 tiles)
            const int split_below = (cfg->geo == G3T && pc.nt >= 4) ? sw.split_t64 : (cfg->geo == G3S1 && pc.nt == 4) || (cfg->geo == G3S2 && pc.nt >= 8) ? sw.split_s64 : ((cfg->geo == G3S1 || cfg->geo == G3T) ? 257 : 256);
            if (t.total_tiles < split_below && pc.nt > 1 && !o.cls && !sw.on(SW_NO_SPLIT)) {
                const int want = (256 + t.total_tiles - 1) / t.total_tiles;   // split factor that would fill the chip (narrow blocks: 512 measured level)
                for (int nts = pc.nt / 2; nts >= 1; nts /= 2) {               // coarsest split first
                    const TileCfg *c2 = tile_cfg_find_like(tp.cfg, nts);
                    if (!c2) continue;
                    cfg = c2;
                    t.nsplit = pc.nt / nts;
                    if (t.nsplit >= want) break;
                }
            }
            t.grid = 8 * ((t.total_tiles + 7) / 8);   // one tile per workgroup, grid a multiple of the 8 XCDs
            // split-K: when even the channel split leaves most CUs idle and the contraction is several channel-group
            // stages deep, the stages are dealt to grid.z workgroups (fp32 partials, summed in fixed order by
            // splitk_finish) so that one workgroup no longer walks all of them in sequence
            t.ksplit = 1;
            t.warm = (t.total_tiles * t.nsplit <= sw.warm_max_wgs) ? 1 : 0;
            float *partial = nullptr;
            const int64_t M_out = (int64_t)out.B * No * Ho * Wo;
            // transposed conv on few tiles: its 4 sub-pixel passes as 4 workgroups (no reduction, any epilogue)
            const int thr = sw.split_wg;
            t.pass_split = (L.transposed && t.total_tiles * t.nsplit <= thr && !sw.on(SW_NO_SPLITK)) ? 1 : 0;
            if (t.pass_split && tp.nstage >= 2 && !sw.on(SW_NO_TEAMS)) {
                // a transposed layer whose passes are workgroups of their own walks its 2-4 channel-group stages one after the other (SPP conv8 at batch 1:
                // 30 us): one stage per team instead, with as few output-channel workgroups per tile as keep the launch to one round of workgroups
                for (int nts = cfg->nt; nts <= pc.nt && nts <= 2; nts *= 2) {
                    const TileCfg *team = tile_cfg_find_team(cfg, tp.nstage, nts);
                    if (!team) continue;
                    const int tz_t = (a.Ng + team->tz - 1) / team->tz, ty_t = (a.Hg + team->ty - 1) / team->ty, tx_t = (a.Wg + team->tx - 1) / team->tx;
                    const int64_t wgs = (int64_t)a.B * tz_t * ty_t * tx_t * (pc.nt / nts) * 4;
                    if (wgs > sw.team_max_wgs) continue;
                    cfg = team;
                    t.nsplit = pc.nt / nts;
                    t.tiles_z = tz_t; t.tiles_y = ty_t; t.tiles_x = tx_t;
                    t.total_tiles = a.B * tz_t * ty_t * tx_t;
                    t.grid = 8 * ((t.total_tiles + 7) / 8);
                    t.warm = (t.total_tiles * t.nsplit <= sw.warm_max_wgs) ? 1 : 0;
                    break;
                }
            }
            if (!t.pass_split && tile_cfg_has_splitk(cfg) && t.total_tiles * t.nsplit <= thr * 3 / 4 && tp.nstage >= 2 && !o.cls && !o.out_pre && !o.outf && !o.discard && !o.res_bcast && L.cout % 4 == 0 &&
                !sw.on(SW_NO_SPLITK)) {
                // enough splits for ~two workgroups per CU (measured 256 ... 768 at batch 1 / 4 and on one End_to_End stack: 512 is 3-4 %
                // faster than the earlier floor(256 / n), which left 129 ... 192-workgroup launches unsplit)
                const int want = (sw.ksplit_target + t.total_tiles * t.nsplit - 1) / (t.total_tiles * t.nsplit);
                t.ksplit = std::max(1, std::min(std::min(tp.nstage, want), 8));
                // the same split INSIDE the workgroup where a team configuration covers it (round 6): the teams' partial sums meet in LDS, no partials through
                // memory and no splitk_finish launch (5-6 us each behind 28 of a batch-1 forward's 89 launches)
                const TileCfg *team = (t.ksplit > 1 && !sw.on(SW_NO_TEAMS)) ? tile_cfg_find_team(cfg, tp.nstage, cfg->nt) : nullptr;
                if (team) {
                    // ... unless the team launch needs several rounds of workgroups (their LDS images allow one or two per CU, and next to the other streams'
                    // kernels they take whole CUs): measured per layer at batch 1 / 2 / 4, profiles/r06_batch1_teams.txt
                    const int tz_t = (a.Ng + team->tz - 1) / team->tz;
                    const int64_t wgs = (int64_t)a.B * tz_t * ((a.Hg + team->ty - 1) / team->ty) * ((a.Wg + team->tx - 1) / team->tx) * t.nsplit;
                    if (wgs < sw.team_min_wgs || wgs > sw.team_max_wgs) team = nullptr;
                }
                if (team) {
                    cfg = team;
                    t.ksplit = 1;
                    t.tiles_z = (a.Ng + cfg->tz - 1) / cfg->tz;
                    t.tiles_y = (a.Hg + cfg->ty - 1) / cfg->ty;
                    t.tiles_x = (a.Wg + cfg->tx - 1) / cfg->tx;
                    t.total_tiles = a.B * t.tiles_z * t.tiles_y * t.tiles_x;
                    t.grid = 8 * ((t.total_tiles + 7) / 8);
                    t.warm = (t.total_tiles * t.nsplit <= sw.warm_max_wgs) ? 1 : 0;
                }
                if (t.ksplit > 1) {
                    t.partial_stride = M_out * (int64_t)pc.nt * 16;
                    partial = (float *)raw(t.ksplit * t.partial_stride * (int64_t)sizeof(float));
                    t.partial = partial;
                    if (!ok()) return out;
                }
            }
            if (dry) {
                drop_raw(partial);
                return out;
            }
            // the pixel-pair stem on whole tiles from the fp32 stack: the persistent pipelined kernel (dffw_stem.hip)
            const bool tracing = sw.trace_layer && sw.trace_out && name == sw.trace_layer;   // stem_pipe has no tile timeline: a traced stem runs on conv_tile
            const bool stem_pipe_run = stem_pair && !sw.on(SW_NO_STEM_PIPE) && !tracing && stem_pipe_ok(e->prec, cfg, a, t);
            auto kernel_name = [&](char *kn, int n) {
                if (stem_pipe_run) return stem_pipe_kernel_name(a, kn, n);
                conv_tile_kernel_name(e->prec, cfg, t.ksplit > 1 || (a.dbg & (DFFW_ARGS_RAW | DFFW_ARGS_SUMS)), tile_lean(e->prec, cfg, a, t), kn, n);
            };
            {
                char kn[96];
                kernel_name(kn, sizeof kn);
                g_last_kernel = kn;
            }
            if (e->profiling) {
                char kn[96];
                kernel_name(kn, sizeof kn);
                const double opx = (double)out.B * No * Ho * Wo;
                const double bytes = (double)in0.pixels() * L.cin * elem_bytes()
                                     + opx * L.cout * (o.outf ? 4.0 : elem_bytes() * (o.out_pre ? 2 : 1))
                                     + opx * L.cout * elem_bytes() * ((o.res0 ? 1 : 0) + (o.res1 ? 1 : 0))
                                     + (double)L.kd * L.kh * L.kw * L.cin * L.cout * elem_bytes();
                prof_begin(kn, name, flops, bytes);
            }
            // debug timeline of one layer: DFFW_TRACE_LAYER=<layer name> DFFW_TRACE_OUT=<file>; per tile 8 x u64
            // (s_memtime at start / fill issued / fill landed / contraction done / stores acknowledged, HW_ID)
            unsigned long long *trace = nullptr;
            const char *tl = sw.trace_layer, *tout = sw.trace_out;
            if (tl && tout && name == tl) {
                check(hipMalloc((void **)&trace, (size_t)t.total_tiles * 64), "trace alloc");
                if (ok()) check(hipMemsetAsync(trace, 0, (size_t)t.total_tiles * 64, s), "trace memset");
                a.trace = trace;
            }
            check(stem_pipe_run ? launch_stem_pipe(a, t, sw.roll_wgs, s) : launch_conv_tile(e->prec, cfg, a, t, s), name.c_str());
            prof_end();
            if (t.ksplit > 1) {
                prof_begin("dffw::splitk_finish_kernel", name + " (split-K finish)", 0.0,
                           (double)M_out * L.cout * (4.0 * t.ksplit + elem_bytes() * (o.res0 ? 2 : 1)));
#ifndef DFFW_EXP_SKIP_FINISH   // (dev-only timing bound, tools/build_variant_lib.sh: what a split-K without its finish launch could save at most; results are garbage)
                check(launch_splitk_finish(e->prec, partial, t.ksplit, t.partial_stride, M_out, pc.nt * 16, L.cout, pc.bias, a.res0,
                                           o.relu, out.p, s), "splitk_finish");
#endif
                prof_end();
                drop_raw(partial);
            }
            if (trace && ok()) {
                std::vector<unsigned long long> host((size_t)t.total_tiles * 8);
                check(hipStreamSynchronize(s), "trace sync");
                check(hipMemcpy(host.data(), trace, host.size() * 8, hipMemcpyDeviceToHost), "trace copy");
                if (FILE *f = fopen(tout, "wb")) {
                    fwrite(host.data(), 8, host.size(), f);
                    fclose(f);
                }
                (void)hipFree(trace);
            }
            return out;
        }
        if (dry) return out;
        for (const Variant &v : pc.variants) {
            a.KC = v.KC;
            a.tab = v.tab;
            a.wpk = v.wpk;
            if (L.transposed) {
                a.Ng = in0.N; a.Hg = in0.H; a.Wg = in0.W;
                a.sy = a.sx = 1;
                a.osy = a.osx = 2;
                a.ooy = v.ooy; a.oox = v.oox;
            } else {
                a.Ng = No; a.Hg = Ho; a.Wg = Wo;
                a.sy = L.sh; a.sx = L.sw;
                a.osy = a.osx = 1;
                a.ooy = a.oox = 0;
            }
            a.M = (int64_t)a.B * a.Ng * a.Hg * a.Wg;
            if (sw.on(SW_NO_SMALL)) a.dbg |= 32;   // (bit 5 of dbg: conv_igemm also for small grids)
            {
                char kn[64];
                conv_kernel_name_for(e->prec, a, kn, sizeof kn);
                g_last_kernel = kn;
            }
            if (e->profiling) {
                char kn[64];
                conv_kernel_name_for(e->prec, a, kn, sizeof kn);
                const double nv = (double)pc.variants.size();
                const double opx = (double)a.M;  // output pixels written by this launch
                double bytes = (double)in0.pixels() * L.cin * elem_bytes() / nv   // input volume read once per layer
                               + opx * L.cout * (o.outf ? 4.0 : elem_bytes() * (o.out_pre ? 2 : 1))
                               + opx * L.cout * elem_bytes() * ((o.res0 ? 1 : 0) + (o.res1 ? 1 : 0))
                               + (double)v.ntaps * L.cin * L.cout * elem_bytes();
                prof_begin(kn, name, 2.0 * (double)a.M * v.ntaps * L.cin * L.cout, bytes);
            }
            check(launch_conv(e->prec, a, s), name.c_str());
            prof_end();
        }
        return out;
    }

    Act pool(const Act &x, int mode, int k) {
        Act out = act(x.B, x.N, x.H / k, x.W / k, x.C);
        if (ok() && !dry) {
            char kn[48];
            snprintf(kn, sizeof kn, "dffw::pool_kernel<%d>", e->prec);
            prof_begin(kn, mode == 0 ? "maxpool" : "avgpool", 0.0, (double)(x.pixels() + out.pixels()) * x.C * elem_bytes());
            check(launch_pool(e->prec, mode, k, x.p, out.p, x.B, x.N, x.H, x.W, x.C, s), "pool");
            prof_end();
        }
        return out;
    }

    void tap(const char *name, const Act &a) {
        if (!ok() || dry) return;
        for (int i = 0; i < n_taps; ++i)
            if (!strcmp(taps[i].name, name)) {
                const int64_t n = a.pixels() * a.C;
                if (taps[i].numel != n) {
                    err = fail(DFFW_EINVAL, "tap %s holds %lld elements, caller gave %lld", name, (long long)n, (long long)taps[i].numel);
                    return;
                }
                check(launch_to_ncdhw(e->prec, a.p, taps[i].dst, a.B, a.C, a.N, a.H, a.W, s), name);
            }
    }
    void tap_f32(const char *name, const float *p, int64_t n) {
        if (!ok() || dry) return;
        for (int i = 0; i < n_taps; ++i)
            if (!strcmp(taps[i].name, name)) {
                if (taps[i].numel != n) {
                    err = fail(DFFW_EINVAL, "tap %s holds %lld elements, caller gave %lld", name, (long long)n, (long long)taps[i].numel);
                    return;
                }
                check(hipMemcpyAsync(taps[i].dst, p, n * sizeof(float), hipMemcpyDeviceToDevice, s), name);
            }
    }
};

// SRD block (DEN.py:317-330): x -> feat = relu(x + BN(conv(relu(BN(conv x))))) ; feat + relu(conv1(relu(conv3x1x1 feat)))
// pooled (optional): receives max_pool(1,2,2) of the block's output when the fused attention kernel can produce it
// on the way (else it is left empty and the caller pools separately).
static Act srd(Run &r, const std::string &p, Act &x, bool drop_x, Act *pooled = nullptr) {
    // the 8-channel block on whole 8 x 16 columns: one fused persistent kernel (dffw_srd_roll.hip)
    {
        auto c0 = r.e->convs.find(p + ".Focus_Measure.conv.0.0"), c2 = r.e->convs.find(p + ".Focus_Measure.conv.2.0");
        auto a3 = r.e->convs.find(p + ".N_ch_attention.0"), a1 = r.e->convs.find(p + ".N_ch_attention.2");
        int sty, stx;
        if (x.C == 16) srd_roll16_tile(&sty, &stx);
        else srd_roll_tile(&sty, &stx);
        const auto end = r.e->convs.end();
        if ((x.C == 8 || x.C == 16) && c0 != end && c2 != end && a3 != end && a1 != end &&
            c0->second.wsrd && c2->second.wsrd && a3->second.w32 &&
            a1->second.w32 && a3->second.watt && a1->second.watt && a3->second.def.kd == 3 && a1->second.def.kd == 1 &&
            x.H % sty == 0 && x.W % stx == 0 && x.H % 2 == 0 && (int64_t)x.B * (x.H / sty) * (x.W / stx) >= r.sw.roll_min_units &&
            !r.sw.on(SW_NO_FUSED_SRD) && !r.sw.on(SW_NO_FUSED_ATTENTION) && !r.sw.on(SW_NO_TILE)) {
            Act out = r.act(x.B, x.N, x.H, x.W, x.C);
            const bool with_pool = pooled && !r.sw.on(SW_NO_FUSED_POOL);
            if (with_pool) *pooled = r.act(x.B, x.N, x.H / 2, x.W / 2, x.C);
            if (r.ok() && !r.dry) {
                if (r.e->ensure_zero_page() != DFFW_OK) { r.err = DFFW_EHIP; return out; }
                SrdArgs a;
                memset(&a, 0, sizeof a);
                a.x = x.p; a.out = out.p; a.pooled = with_pool ? pooled->p : nullptr;
                a.w0 = c0->second.wsrd; a.w2 = c2->second.wsrd;
                a.b0 = c0->second.bias; a.b2 = c2->second.bias;
                a.w3 = a3->second.w32; a.w1 = a1->second.w32;
                a.w3f = a3->second.watt; a.w1f = a1->second.watt;
                a.zero = r.e->zero_page;
                a.B = x.B; a.N = x.N; a.H = x.H; a.W = x.W;
                a.tiles_y = x.H / sty; a.tiles_x = x.W / stx;
                a.total_tiles = x.B * a.tiles_y * a.tiles_x;
                a.wgs = r.sw.srd_wgs;
                char kn[64];
                const bool pipe16 = x.C == 16 && r.sw.srd_pipe;   // (opt-in: measured 6 % slower than srd_roll16, profiles/r06_srd_two_slice.txt)
                if (pipe16) srd_pipe16_kernel_name(r.e->prec, with_pool, kn, sizeof kn);
                else if (x.C == 16) srd_roll16_kernel_name(r.e->prec, with_pool, kn, sizeof kn);
                else srd_roll_kernel_name(r.e->prec, with_pool, kn, sizeof kn);
                g_last_kernel = kn;
                const double px = (double)x.pixels();
                // algorithmic: two 1x3x3 C -> C convs + the 3x1x1 and 1x1x1 attention convs; x read once, out (+ pooled) written once
                r.prof_begin(kn, p, 2.0 * px * (2 * 9 + 4) * x.C * x.C, (with_pool ? 2.25 : 2.0) * px * x.C * r.elem_bytes());
#ifdef DFFW_TRACE_BUILD
                a.trace = r.trace_begin(p, 1024, 4);
#endif
                r.check(pipe16 ? launch_srd_pipe16(r.e->prec, a, r.s) : x.C == 16 ? launch_srd_roll16(r.e->prec, a, r.s) : launch_srd_roll(r.e->prec, a, r.s), "srd_roll");
                r.prof_end();
                r.trace_end();
            }
            if (drop_x) r.drop(x);
            return out;
        }
    }
    ConvOpt o1; o1.relu = 1;
    Act t = r.conv(p + ".Focus_Measure.conv.0.0", x, o1);
    ConvOpt o2; o2.relu = 1; o2.res0 = &x;
    Act feat = r.conv(p + ".Focus_Measure.conv.2.0", t, o2);
    r.drop(t);
    if (drop_x) r.drop(x);
    Act out;
    auto i3 = r.e->convs.find(p + ".N_ch_attention.0");
    auto i1 = r.e->convs.find(p + ".N_ch_attention.2");
    if (srd_attention_supported(feat.C) && i3 != r.e->convs.end() && i1 != r.e->convs.end() && i3->second.w32 && i1->second.w32 &&
        !r.sw.on(SW_NO_FUSED_ATTENTION)) {
        out = r.act(feat.B, feat.N, feat.H, feat.W, feat.C);
        const bool with_pool = pooled && !r.sw.on(SW_NO_FUSED_POOL);
        if (with_pool) *pooled = r.act(feat.B, feat.N, feat.H / 2, feat.W / 2, feat.C);
        if (r.ok() && !r.dry) {
            char kn[64];
            snprintf(kn, sizeof kn, "dffw::srd_attention_kernel<%d, %d>", r.e->prec, feat.C);
            const double px = (double)feat.pixels();
            r.prof_begin(kn, p + ".N_ch_attention", 2.0 * px * 4 * feat.C * feat.C, (with_pool ? 2.25 : 2.0) * px * feat.C * r.elem_bytes());
            r.check(launch_srd_attention(r.e->prec, feat.p, out.p, i3->second.w32, i1->second.w32, feat.B, feat.N, feat.H, feat.W,
                                         feat.C, with_pool ? pooled->p : nullptr, r.s), "srd_attention");
            r.prof_end();
        }
    } else if (feat.C == 32 && i3 != r.e->convs.end() && i1 != r.e->convs.end() && i3->second.watt && i1->second.watt && feat.W % 16 == 0 &&
               !r.sw.on(SW_NO_FUSED_ATTENTION)) {
        out = r.act(feat.B, feat.N, feat.H, feat.W, feat.C);
        if (r.ok() && !r.dry) {
            char kn[64];
            snprintf(kn, sizeof kn, "dffw::srd_attention_mfma<%d>", r.e->prec);
            const double px = (double)feat.pixels();
            r.prof_begin(kn, p + ".N_ch_attention", 2.0 * px * 4 * feat.C * feat.C, 2.0 * px * feat.C * r.elem_bytes());
            r.check(launch_srd_attention_mfma(r.e->prec, feat.p, out.p, i3->second.watt, i1->second.watt, feat.B, feat.N, feat.H, feat.W, r.s),
                    "srd_attention_mfma");
            r.prof_end();
        }
    } else {
        ConvOpt o3; o3.relu = 1;
        Act a = r.conv(p + ".N_ch_attention.0", feat, o3);
        ConvOpt o4; o4.relu = 2; o4.res0 = &feat;
        out = r.conv(p + ".N_ch_attention.2", a, o4);
        r.drop(a);
    }
    r.drop(feat);
    return out;
}

// EFD block (DEN.py:306-315)
static Act efd(Run &r, const std::string &p, const Act &x, Act *pooled = nullptr) {
    // the 8 -> 16 channel block with its pooled input at hand: both branches in one rolling kernel (conv_roll_efd)
    {
        auto ca = r.e->convs.find(p + ".stride_conv.0"), cb = r.e->convs.find(p + ".max_pooling.1.0");
        int ty, tx;
        efd_roll_tile(&ty, &tx);
        const int Ho = x.H / 2, Wo = x.W / 2;
        const auto end = r.e->convs.end();
        if (x.C == 8 && pooled && pooled->p && ca != end && cb != end && ca->second.wroll8 && cb->second.wroll8 && x.H % 2 == 0 && x.W % 2 == 0 &&
            Ho % ty == 0 && Wo % tx == 0 && (int64_t)x.B * (Ho / ty) * (Wo / tx) >= r.sw.roll_min_units && !r.sw.on(SW_NO_ROLL) &&
            !r.sw.on(SW_NO_FUSED_EFD) && !r.sw.on(SW_NO_TILE)) {
            Act out = r.act(x.B, x.N, Ho, Wo, 16);
            if (r.ok() && !r.dry) {
                if (r.e->ensure_zero_page() != DFFW_OK) { r.err = DFFW_EHIP; return out; }
                ConvArgs a;
                memset(&a, 0, sizeof a);
                a.in0 = x.p; a.C0 = 8;
                a.in1 = pooled->p; a.C1 = 8;
                a.B = x.B; a.Ni = x.N; a.Hi = x.H; a.Wi = x.W;
                a.Ng = x.N; a.Hg = Ho; a.Wg = Wo;
                a.No = x.N; a.Ho = Ho; a.Wo = Wo;
                a.Cout = 16;
                a.bias = ca->second.bias;
                a.out = out.p;
                a.relu = 1;
                a.zero = r.e->zero_page;
                a.M = (int64_t)x.B * x.N * Ho * Wo;
                a.dbg = (r.sw.debug_flags & 6) | r.sw.path_bits();
                RollArgs t;
                memset(&t, 0, sizeof t);
                t.wroll = ca->second.wroll8;
                t.wroll2 = cb->second.wroll8;
                t.bias2 = cb->second.bias;
                t.tiles_y = Ho / ty; t.tiles_x = Wo / tx;
                t.zsplit = 1;
                t.total_tiles = x.B * t.tiles_y * t.tiles_x;
                t.wgs = r.sw.roll_wgs;
                char kn[64];
                conv_roll_efd_kernel_name(r.e->prec, a, true, kn, sizeof kn);
                g_last_kernel = kn;
                const double opx = (double)x.B * x.N * Ho * Wo;
                r.prof_begin(kn, p, 2.0 * opx * 27.0 * 8 * 16 * 2, ((double)x.pixels() * 8 + opx * 8 + opx * 16) * r.elem_bytes());
                r.check(launch_conv_roll_efd(r.e->prec, a, t, r.s), "conv_roll_efd");
                r.prof_end();
            }
            r.drop(*pooled);
            return out;
        }
    }
    // the 16 -> 32 channel block (`FM_conv2.0`): both branches in one streaming kernel, the waves split by branch / output tile / pixel half (conv_efd16)
    {
        auto ca = r.e->convs.find(p + ".stride_conv.0"), cb = r.e->convs.find(p + ".max_pooling.1.0");
        int ty, tx;
        efd16_tile(&ty, &tx);
        const int Ho = x.H / 2, Wo = x.W / 2;
        const auto end = r.e->convs.end();
        if (x.C == 16 && pooled && pooled->p && ca != end && cb != end && ca->second.wroll_s2 && cb->second.wroll15 && ca->second.def.cout == 32 && x.H % 2 == 0 &&
            x.W % 2 == 0 && Ho % ty == 0 && Wo % tx == 0 && (int64_t)x.B * (Ho / ty) * (Wo / tx) >= r.sw.roll_min_units && !r.sw.on(SW_NO_ROLL) &&
            !r.sw.on(SW_NO_FUSED_EFD) && !r.sw.on(SW_NO_TILE)) {
            ConvArgs a;
            memset(&a, 0, sizeof a);
            a.in0 = x.p; a.C0 = 16;
            a.in1 = pooled->p; a.C1 = 16;
            a.B = x.B; a.Ni = x.N; a.Hi = x.H; a.Wi = x.W;
            a.Ng = x.N; a.Hg = Ho; a.Wg = Wo;
            a.No = x.N; a.Ho = Ho; a.Wo = Wo;
            a.Cout = 32;
            a.bias = ca->second.bias;
            a.relu = 1;
            a.M = (int64_t)x.B * x.N * Ho * Wo;
            RollArgs t;
            memset(&t, 0, sizeof t);
            t.wroll = ca->second.wroll_s2;
            t.wroll2 = cb->second.wroll15;
            t.bias2 = cb->second.bias;
            t.tiles_y = Ho / ty; t.tiles_x = Wo / tx;
            t.zsplit = 1;
            t.total_tiles = x.B * t.tiles_y * t.tiles_x;
            t.wgs = r.sw.roll_wgs;
            a.out = (uint16_t *)16;   // (placeholder for the check below: the output is allocated once the kernel is known to serve the shape)
            if (efd16_ok(r.e->prec, a, t)) {
                Act out = r.act(x.B, x.N, Ho, Wo, 32);
                if (r.ok() && !r.dry) {
                    a.out = out.p;
                    char kn[64];
                    conv_efd16_kernel_name(kn, sizeof kn);
                    g_last_kernel = kn;
                    const double opx = (double)x.B * x.N * Ho * Wo;
                    r.prof_begin(kn, p, 2.0 * opx * 27.0 * 16 * 32 * 2, ((double)x.pixels() * 16 + opx * 16 + opx * 32) * r.elem_bytes());
                    r.check(launch_conv_efd16(a, t, r.s), "conv_efd16");
                    r.prof_end();
                }
                r.drop(*pooled);
                return out;
            }
        }
    }
    Act a = r.conv(p + ".stride_conv.0", x);
    Act m = (pooled && pooled->p) ? *pooled : r.pool(x, 0, 2);
    ConvOpt o; o.relu = 1; o.res0 = &a;
    Act out = r.conv(p + ".max_pooling.1.0", m, o);
    r.drop(m);
    r.drop(a);
    return out;
}

// one scale of the pyramid: r0 = dresX_0(t) ; dresX_1(r0) + r0   (DEN.py:216-223)
static Act pyramid_scale(Run &r, const std::string &S, const char *tag, Act &t) {
    const std::string d = S + ".dres" + tag;
    ConvOpt rl; rl.relu = 1;
    Act a = r.conv(d + "_0.0.0", t, rl);
    Act r0 = r.conv(d + "_0.2.0", a, rl);
    r.drop(a);
    Act b = r.conv(d + "_1.0.0", r0, rl);
    ConvOpt o; o.res0 = &r0;
    Act out = r.conv(d + "_1.2.0", b, o);
    r.drop(b);
    r.drop(r0);
    return out;
}

// hourglassup.forward (DEN.py:212-238)
static Act pyramid(Run &r, const std::string &S, const Act &v3) {
    ConvOpt rl; rl.relu = 1;
    Act p8, p16, p32;
    if (v3.H % 8 == 0 && v3.W % 8 == 0 && !r.sw.on(SW_NO_POOL3)) {   // one pass over v3 for the three pyramid scales (pool3_kernel)
        p8 = r.act(v3.B, v3.N, v3.H / 2, v3.W / 2, v3.C);
        p16 = r.act(v3.B, v3.N, v3.H / 4, v3.W / 4, v3.C);
        p32 = r.act(v3.B, v3.N, v3.H / 8, v3.W / 8, v3.C);
        if (r.ok() && !r.dry) {
            char kn[48];
            snprintf(kn, sizeof kn, "dffw::pool3_kernel<%d>", r.e->prec);
            r.prof_begin(kn, "avgpool (1,2,2)+(1,4,4)+(1,8,8)", 0.0, (double)(v3.pixels() + p8.pixels() + p16.pixels() + p32.pixels()) * v3.C * r.elem_bytes());
            r.check(launch_pool3(r.e->prec, v3.p, p8.p, p16.p, p32.p, v3.B, v3.N, v3.H, v3.W, v3.C, r.s), "pool3");
            r.prof_end();
        }
    } else {
        p8 = r.pool(v3, 1, 2);
        p16 = r.pool(v3, 1, 4);
        p32 = r.pool(v3, 1, 8);
    }
    // the three scales are independent chains of 4 convs (DEN.py:216-223): side by side when concurrency is on
    r.forked = r.concurrent;
    // (one hipEventRecord per side stream: with ONE event that both side streams wait for the batch-1 forward is 80 us SLOWER, 0.877 -> 0.958 ms; and
    // chaining the joins -- side 1 waits for side 0, the main stream for side 1 -- costs as much: profiles/r06_batch1_teams.txt)
    r.fork(0);
    r.fork(1);
    Act s8 = pyramid_scale(r, S, "8", p8);
    r.drop(p8);
    r.on(0);
    Act s16 = pyramid_scale(r, S, "16", p16);
    r.drop(p16);
    r.on(1);
    Act s32 = pyramid_scale(r, S, "32", p32);
    r.drop(p32);
    r.on(-1);
    r.join(0);
    r.join(1);
    r.forked = false;
    r.release_deferred();
    // redir1 / redir2 (1x1x1 conv + BN of x_8 / of conv2's output, DEN.py:209-210,234-237) only feed the residual inputs of conv9 / conv8: they run on the
    // side streams next to the chain conv1 ... conv4 instead of between its launches (two small gather-GEMM launches off the critical path)
    // (below 2M stack pixels -- batch 1 and 2 of 10x256x256 -- in line: a fork and a join hold the main queue for ~6 us each, the launch itself is 7-9 us:
    // 0.877 -> 0.870 ms at batch 1, 1.122 -> 1.118 at batch 2; from batch 4 up the side streams win by 0.3-0.7 %)
    const bool redir_side = r.concurrent && (r.sw.redir_side == 1 || (r.sw.redir_side < 0 && (int64_t)v3.B * v3.N * v3.H * v3.W * 16 >= (2 << 20)));
    r.forked = redir_side;
    if (redir_side) {
        r.fork(0);
        r.on(0);
    }
    Act rd1 = r.conv(S + ".redir1.0", s8);
    r.on(-1);
    Act d1 = r.conv(S + ".conv1", s8);
    r.drop(s8);
    ConvOpt c1 = rl; c1.in1 = &s16;
    Act m1 = r.conv(S + ".combine1.0.0", d1, c1);
    r.drop(d1); r.drop(s16);
    Act c2 = r.conv(S + ".conv2.0.0", m1, rl);
    r.drop(m1);
    if (redir_side) {
        r.fork(1);
        r.on(1);
    }
    Act rd2 = r.conv(S + ".redir2.0", c2);
    r.on(-1);
    Act d2 = r.conv(S + ".conv3", c2);
    r.drop(c2);
    ConvOpt cc2 = rl; cc2.in1 = &s32;
    Act m2 = r.conv(S + ".combine2.0.0", d2, cc2);
    r.drop(d2); r.drop(s32);
    Act c4 = r.conv(S + ".conv4.0.0", m2, rl);
    r.drop(m2);
    if (redir_side) r.join(1);
    ConvOpt u8o = rl; u8o.res0 = &rd2;
    Act u8 = r.conv(S + ".conv8.0", c4, u8o);
    r.drop(c4); r.drop(rd2);
    if (redir_side) r.join(0);
    ConvOpt u9o = rl; u9o.res0 = &rd1;
    Act u9 = r.conv(S + ".conv9.0", u8, u9o);
    r.drop(u8); r.drop(rd1);
    r.forked = false;
    r.release_deferred();
    return u9;
}

// hourglass.forward (DEN.py:265-284).  x = cat[xa, xb] on channels.  Returns conv6's output `out`
// in *out_raw (if wanted) and out + skip in the return value; pre1 = conv0's output.
static Act hourglass(Run &r, const std::string &p, const Act &xa, const Act &xb, const Act *presqu, const Act *postsqu,
                     const Act &skip, Act *pre1_out, Act *out_raw, const std::string &cls, float *cls_out, bool discard_sum) {
    ConvOpt rl; rl.relu = 1;
    ConvOpt c0 = rl; c0.in1 = &xb;
    Act pre1 = r.conv(p + ".conv0.0.0", xa, c0);
    Act o1 = r.conv(p + ".conv1.0.0", pre1, rl);
    ConvOpt c2 = rl; c2.res0 = postsqu;
    Act pre = r.conv(p + ".conv2.0", o1, c2);
    r.drop(o1);
    Act o3 = r.conv(p + ".conv3.0.0", pre, rl);
    Act o4 = r.conv(p + ".conv4.0.0", o3, rl);
    r.drop(o3);
    ConvOpt c5 = rl; c5.res0 = presqu ? presqu : &pre;
    Act o5 = r.conv(p + ".conv5.0", o4, c5);
    r.drop(o4); r.drop(pre);
    // conv6 + skip (DEN.py:96,102,107) with the 1x1x1 classifier (DEN.py:97,103,108) folded into the epilogue
    ConvOpt c6; c6.res0 = &skip; c6.out_pre = out_raw; c6.cls = cls.c_str(); c6.cls_out = cls_out; c6.discard = discard_sum;
    Act sum = r.conv(p + ".conv6.0", o5, c6);
    r.drop(o5);
    if (pre1_out) *pre1_out = pre1; else r.drop(pre1);
    return sum;
}

// The four regression heads (mid_out, pred1..3) run as ONE launch at the end of the forward (DFFW_NO_REGRESS_MERGE: one launch each,
// where the score volume is ready): a head is queued here, its score volume stays allocated until flush_regress.
struct RegressQueue {
    RegressHeads hd{};
    void *keep[4] = {nullptr, nullptr, nullptr, nullptr};
    int nkeep = 0;
    double bytes = 0.0;
};
static void regress(Run &r, RegressQueue &q, const char *tag, float *score, int B, int N, int h, int w, int H, int W, const float *fd,
                    const int64_t fst[4], float *out, bool merge) {
    if (!merge) {
        if (r.ok() && !r.dry && out) {
            r.prof_begin("dffw::regress_kernel", tag, 0.0, (double)B * N * h * w * 4.0 + (double)B * H * W * 4.0);
            r.check(launch_regress(score, B, N, h, w, H, W, fd, fst[0], fst[1], fst[2], fst[3], out, r.s), tag);
            r.prof_end();
        }
        r.drop_raw(score);
        return;
    }
    // (the score volume is kept in the dry run that sizes the workspace exactly as in the real one)
    q.keep[q.nkeep++] = score;
    if (!out || r.dry) return;
    const int k = q.hd.n++;
    q.hd.score[k] = score;
    q.hd.depth[k] = out;
    q.hd.h[k] = h;
    q.hd.w[k] = w;
    q.bytes += (double)B * N * h * w * 4.0 + (double)B * H * W * 4.0;
}
static void flush_regress(Run &r, RegressQueue &q, int B, int N, int H, int W, const float *fd, const int64_t fst[4]) {
    if (q.hd.n && r.ok() && !r.dry) {
        q.hd.nofuse = r.sw.on(SW_NO_REGRESS_FUSED) ? 1 : 0;
        const bool fused = q.hd.n > 1 && N <= 16 && (int64_t)B * H * W < (1ll << 31) && !q.hd.nofuse;   // launch_regress_heads' own test
        // algorithmic bytes: the score volumes and depth maps + a dense focus-distance map once (the fused kernel does read it once; per-head launches re-read it)
        const double bytes = q.bytes + ((fst[2] || fst[3]) ? (double)B * N * H * W * 4.0 : 0.0);
        r.prof_begin(fused ? "dffw::regress_fused_kernel" : "dffw::regress_kernel", "regress.mid_out+pred1+pred2+pred3", 0.0, bytes);
        r.check(launch_regress_heads(q.hd, B, N, H, W, fd, fst[0], fst[1], fst[2], fst[3], r.s), "regress");
        r.prof_end();
    }
    for (int k = 0; k < q.nkeep; ++k) r.drop_raw(q.keep[k]);
    q.hd.n = 0;
    q.nkeep = 0;
}

// raw != null: FS is not given; the stem reads the raw stack (or, when the tiled stem kernel does not serve this
// shape, the stack is first expanded into a temporary fp32 volume from the workspace)
static int run_depth(Run &r, const float *FS, const float *fd, const int64_t fst[4], int B, int N, int H, int W, float *const out[4],
                     const RawStack *raw = nullptr) {
    const std::string P = "DFF_net";
    const int prec = r.e->prec;
    ConvOpt rl; rl.relu = 1;
    // the low-resolution layers of the pyramid cannot fill 256 CUs on their own (at batch 32 its 1/32-resolution scale is
    // 128 tiles per launch): the three pyramid scales run side by side on the main + two side streams (measured +7 % at
    // batch 1, +5 % at batch 4, +4 % at batch 8, +1..2.5 % at batch 32; the regression heads on a side stream gained
    // nothing).  DFFW_CONCURRENT_MAX_PIXELS restricts it to stacks below that many pixels.
    {
        const int64_t z = r.sw.concurrent_max_pixels;
        // (not in profiling mode: the per-launch event durations are meant to be each kernel's own)
        const int64_t px = (int64_t)B * N * H * W;
        if ((z < 0 || px < z) && px >= r.sw.concurrent_min_pixels && !r.sw.on(SW_NO_CONCURRENT) && !r.e->profiling) r.enable_concurrency();
    }

    // feature extraction: V1 (8ch, full), V2 (16ch, 1/2), V3 (32ch, 1/4)            DEN.py:77-80
    const std::string stem_name = P + ".FM_measure.Focus_extraction.0.0";
    Act stem;
    if (r.tiled(stem_name, H, W) && !r.sw.on(SW_NO_FUSED_STEM)) {
        // the tiled stem kernel builds its paired-pixel records from the fp32 stack on the fly: no record volume
        Act geom;
        geom.B = B; geom.N = N; geom.H = H; geom.W = W + 2; geom.C = 8;
        ConvOpt so = rl;
        so.fs32 = FS;
        RawStack *rdev = nullptr;
        if (raw) {
            rdev = (RawStack *)r.raw(256);
            if (r.ok() && !r.dry) r.check(launch_set_raw(*raw, rdev, r.s), "set_raw");
            so.fs32 = (const float *)rdev;
            so.raw = true;
        }
        stem = r.conv(stem_name, geom, so);
        r.drop_raw(rdev);
    } else {
        float *tmp = nullptr;
        if (raw) {
            tmp = (float *)r.raw((int64_t)B * 3 * N * H * W * (int64_t)sizeof(float));
            if (r.ok() && !r.dry) {
                const int64_t st[5] = {raw->sb, raw->sn, raw->sy, raw->sx, raw->sc};
                if (dffw_pack_stack(r.e->device, raw->p, raw->dtype, st, B, N, raw->h, raw->w, H, W, tmp, r.s) != DFFW_OK) r.err = DFFW_EHIP;
            }
            FS = tmp;
        }
        Act in = r.act(B, N, H, W + 2, 8);   // paired-pixel records, see stack_in_kernel
        if (r.ok() && !r.dry) {
            char kn[48];
            snprintf(kn, sizeof kn, "dffw::stack_in_kernel<%d>", prec);
            r.prof_begin(kn, "stack_in", 0.0, (double)B * N * H * W * 3 * 4.0 + (double)B * N * H * (W + 2) * 8 * r.elem_bytes());
            r.check(launch_stack_in(prec, FS, in.p, B, N, H, W, r.s), "stack_in");
            r.prof_end();
        }
        stem = r.conv(stem_name, in, rl);
        r.drop(in);
        r.drop_raw(tmp);
    }
    Act v1p, v2p;   // max-pooled copies written by the attention kernels on the way (EFD's second branch)
    Act v1 = srd(r, P + ".FM_measure.Focus_extraction.2", stem, true, &v1p);
    r.tap("V1", v1);
    Act e1 = efd(r, P + ".FM_conv1.0", v1, &v1p);
    Act v2 = srd(r, P + ".FM_conv1.1", e1, true, &v2p);
    r.tap("V2", v2);
    Act e2 = efd(r, P + ".FM_conv2.0", v2, &v2p);
    Act v3 = srd(r, P + ".FM_conv2.1", e2, true);
    r.tap("V3", v3);

    // multi-scale aggregation (1st hourglass)                                      DEN.py:82
    Act vol = pyramid(r, P + ".SPP_module", v3);
    r.tap("FS_volume", vol);

    // confidence head -> mid_out                                                   DEN.py:83-90
    // (on side stream 0 next to dres0 / deconv_1 below 16M stack pixels: two 1/8-resolution convs and a regression head that
    // nothing else waits for -- measured +2.7 % at batch 1, +2.3 % at batch 8, -0.3 % at batch 32 where dres0 fills the chip)
    RegressQueue rq;
    const bool merge_heads = true;   // (DFFW_NO_REGRESS_MERGE retired in round 5: one launch per head lost the A/B of rounds 3 and 4)
    const int h8 = H / 8, w8 = W / 8;
    float *conf = (float *)r.raw((int64_t)B * N * h8 * w8 * sizeof(float));
    const bool conf_side = r.concurrent && (int64_t)B * N * H * W < (16 << 20) && !r.sw.on(SW_NO_CONF_FORK);
    r.forked = conf_side;
    if (conf_side) {
        r.fork(0);
        r.on(0);
    }
    {
        Act c = r.conv(P + ".confidence.0.0", vol, rl);
        ConvOpt of; of.outf = conf;
        r.conv(P + ".confidence.2", c, of);
        r.drop(c);
        r.tap_f32("conf", conf, (int64_t)B * N * h8 * w8);
        regress(r, rq, "regress.mid_out", conf, B, N, h8, w8, H, W, fd, fst, out[0], merge_heads);
    }
    r.on(-1);

    // refinement                                                                   DEN.py:92-108
    Act d0 = r.conv(P + ".dres0.0.0", vol, rl);
    r.drop(vol);
    Act d1 = r.conv(P + ".dres0.2.0", d0, rl);
    r.drop(d0);
    Act x1 = r.conv(P + ".deconv_1.0", d1);
    r.drop(d1);
    if (conf_side) r.join(0);
    r.forked = false;
    r.release_deferred();

    Act pre_a, out_a;
    const int h4 = H / 4, w4 = W / 4;
    float *cost1 = (float *)r.raw((int64_t)B * N * h4 * w4 * sizeof(float));
    Act s1 = hourglass(r, P + ".dres2", x1, v3, nullptr, nullptr, x1, &pre_a, &out_a, P + ".classif1.0", cost1, false);
    r.drop(x1); r.drop(v3);
    r.tap_f32("cost1", cost1, (int64_t)B * N * h4 * w4);
    regress(r, rq, "regress.pred1", cost1, B, N, h4, w4, H, W, fd, fst, out[1], merge_heads);

    Act x2 = r.conv(P + ".deconv_2.0", s1);
    r.drop(s1);
    Act pre_b, out_b;
    const int h2 = H / 2, w2 = W / 2;
    float *cost2 = (float *)r.raw((int64_t)B * N * h2 * w2 * sizeof(float));
    Act s2 = hourglass(r, P + ".dres3", x2, v2, &pre_a, &out_a, x2, &pre_b, &out_b, P + ".classif2.0", cost2, false);
    r.drop(x2); r.drop(v2); r.drop(pre_a); r.drop(out_a);
    r.tap_f32("cost2", cost2, (int64_t)B * N * h2 * w2);
    regress(r, rq, "regress.pred2", cost2, B, N, h2, w2, H, W, fd, fst, out[2], merge_heads);

    Act x3 = r.conv(P + ".deconv_3.0", s2);
    r.drop(s2);
    float *cost3 = (float *)r.raw((int64_t)B * N * H * W * sizeof(float));
    Act s3 = hourglass(r, P + ".dres4", x3, v1, &pre_b, &out_b, x3, nullptr, nullptr, P + ".classif3.0", cost3, true);  // only its score is used
    r.drop(x3); r.drop(v1); r.drop(pre_b); r.drop(out_b);
    r.drop(s3);
    r.tap_f32("cost3", cost3, (int64_t)B * N * H * W);
    regress(r, rq, "regress.pred3", cost3, B, N, H, W, H, W, fd, fst, out[3], merge_heads);
    flush_regress(r, rq, B, N, H, W, fd, fst);
    return r.err;
}

// resnet_block_2d_OF (End_to_End.py:135-145): relu(feature(x) + BN(conv(relu(BN(conv_s(x))))))
static Act of_block(Run &r, const std::string &p, const Act &x) {
    ConvOpt rl; rl.relu = 1;
    // stride-1 block with 16 output channels on whole 8 x 16 columns: conv.0, conv.2 and the shortcut in one streaming kernel
    {
        auto c0 = r.e->convs.find(p + ".conv.0.0"), c2 = r.e->convs.find(p + ".conv.2.0");
        const auto end = r.e->convs.end();
        const int co = (c0 != end) ? c0->second.def.cout : 0;
        if (r.e->convs.find(p + ".feature") == end && c0 != end && c2 != end && c0->second.wsrd && c2->second.wsrd && (x.C == 8 || x.C == 16) &&
            (co == 16 || (co == 8 && x.C == 8)) && c2->second.def.cout == co && c2->second.cin_all == co + x.C && x.H % 8 == 0 && x.W % 16 == 0 &&
            (int64_t)x.B * (x.H / 8) * (x.W / 16) >= r.sw.roll_min_units && !r.sw.on(SW_NO_FUSED_OF) && !r.sw.on(SW_NO_TILE)) {
            Act out = r.act(x.B, x.N, x.H, x.W, co);
            if (r.ok() && !r.dry) {
                if (r.e->ensure_zero_page() != DFFW_OK) { r.err = DFFW_EHIP; return out; }
                SrdArgs a;
                memset(&a, 0, sizeof a);
                a.x = x.p; a.out = out.p;
                a.w0 = c0->second.wsrd; a.w2 = c2->second.wsrd;
                a.b0 = c0->second.bias; a.b2 = c2->second.bias;
                a.zero = r.e->zero_page;
                a.B = x.B; a.N = x.N; a.H = x.H; a.W = x.W;
                a.tiles_y = x.H / 8; a.tiles_x = x.W / 16;
                a.total_tiles = x.B * a.tiles_y * a.tiles_x;
                a.wgs = r.sw.srd_wgs;
                char kn[64];
                if (co == 8) of_roll8_kernel_name(r.e->prec, kn, sizeof kn);
                else of_roll_kernel_name(r.e->prec, x.C == 8, kn, sizeof kn);
                g_last_kernel = kn;
                const double px = (double)x.pixels();
                const LayerDef &L0 = c0->second.def;
                r.prof_begin(kn, p, 2.0 * px * (9.0 * L0.cin * co + 9.0 * co * co + (double)L0.cin * co), px * (x.C + co) * r.elem_bytes());
                r.check(co == 8 ? launch_of_roll8(r.e->prec, a, r.s) : launch_of_roll(r.e->prec, x.C == 8, a, r.s), "of_roll");
                r.prof_end();
            }
            return out;
        }
    }
    {
        // the 8 -> 16 down-sampling block on whole 8 x 16 output columns: one streaming kernel (of_s2_kernel, dffw_srd_roll.hip)
        auto c0 = r.e->convs.find(p + ".conv.0.0"), c2 = r.e->convs.find(p + ".conv.2.0"), cf = r.e->convs.find(p + ".feature");
        const auto end = r.e->convs.end();
        if (c0 != end && c2 != end && cf != end && x.C == 8 && c0->second.wsrd && c2->second.wsrd && cf->second.wsrd && c0->second.def.sh == 2 &&
            c0->second.def.cout == 16 && c2->second.def.cout == 16 && c2->second.cin_all == 16 && cf->second.def.sh == 2 && cf->second.def.cout == 16 &&
            x.H % 16 == 0 && x.W % 32 == 0 && (int64_t)x.B * (x.H / 16) * (x.W / 32) >= r.sw.roll_min_units && !r.sw.on(SW_NO_FUSED_OF) && !r.sw.on(SW_NO_TILE)) {
            Act out = r.act(x.B, x.N, x.H / 2, x.W / 2, 16);
            if (r.ok() && !r.dry) {
                SrdArgs a;
                memset(&a, 0, sizeof a);
                a.x = x.p; a.out = out.p;
                a.w0 = c0->second.wsrd; a.w2 = c2->second.wsrd; a.w3f = cf->second.wsrd;
                a.b0 = c0->second.bias; a.b2 = c2->second.bias;
                a.B = x.B; a.N = x.N; a.H = out.H; a.W = out.W;
                a.tiles_y = out.H / 8; a.tiles_x = out.W / 16;
                a.total_tiles = x.B * a.tiles_y * a.tiles_x;
                a.wgs = r.sw.srd_wgs;
                char kn[64];
                of_s2_kernel_name(r.e->prec, kn, sizeof kn);
                g_last_kernel = kn;
                const double px = (double)out.pixels();
                r.prof_begin(kn, p, 2.0 * px * (9.0 * 8 * 16 + 9.0 * 16 * 16 + 8.0 * 16), (4.0 * px * 8 + px * 16) * r.elem_bytes());
                r.check(launch_of_s2(r.e->prec, a, r.s), "of_s2");
                r.prof_end();
            }
            return out;
        }
    }
    Act t = r.conv(p + ".conv.0.0", x, rl);
    if (r.e->convs.find(p + ".feature") == r.e->convs.end()) {   // stride-1 block: shortcut folded into conv.2 over [t | x]
        ConvOpt o; o.relu = 1; o.in1 = &x;
        Act out = r.conv(p + ".conv.2.0", t, o);
        r.drop(t);
        return out;
    }
    Act f = r.conv(p + ".feature", x);
    ConvOpt o; o.relu = 1; o.res0 = &f;
    Act out = r.conv(p + ".conv.2.0", t, o);
    r.drop(t);
    r.drop(f);
    return out;
}

// End_to_End.Network.forward (End_to_End.py:13-16): FlowNetwork.forward (End_to_End.py:71-105) aligns the stack,
// DFF_net runs on the aligned stack.  `aligned` receives the warped focal stack (the 5th return value).
static int run_e2e(Run &r, const float *FS, const float *fd, const int64_t fst[4], const float *fov, int B, int N, int H, int W,
                   float *const out[4], float *aligned) {
    const std::string P = "optical_flow_aggregation";
    const int prec = r.e->prec;
    ConvOpt rl; rl.relu = 1;
    // three feature levels: full, 1/2, 1/4 resolution                               End_to_End.py:72-74
    Act a0;
    {
        // the first block reads the fp32 stack itself when its streaming kernel applies (of_first_kernel = of_roll8 with the record
        // conversion inside its fill): the 8-channel record volume of the stack is neither written nor read
        const std::string p0 = P + ".OF_feature.0";
        auto c0 = r.e->convs.find(p0 + ".conv.0.0"), c2 = r.e->convs.find(p0 + ".conv.2.0");
        const auto end = r.e->convs.end();
        const bool first = c0 != end && c2 != end && r.e->convs.find(p0 + ".feature") == end && c0->second.wsrd && c2->second.wsrd &&
                           c0->second.def.cin == 3 && c0->second.def.cout == 8 && c2->second.def.cout == 8 && c2->second.cin_all == 16 && H % 8 == 0 &&
                           W % 16 == 0 && (int64_t)B * (H / 8) * (W / 16) >= r.sw.roll_min_units && !r.sw.on(SW_NO_FUSED_OF) && !r.sw.on(SW_NO_OF_FIRST) &&
                           !r.sw.on(SW_NO_TILE);
        if (first) {
            a0 = r.act(B, N, H, W, 8);
            if (r.ok() && !r.dry) {
                SrdArgs a;
                memset(&a, 0, sizeof a);
                a.w3 = FS; a.out = a0.p;
                a.w0 = c0->second.wsrd; a.w2 = c2->second.wsrd;
                a.b0 = c0->second.bias; a.b2 = c2->second.bias;
                a.B = B; a.N = N; a.H = H; a.W = W;
                a.tiles_y = H / 8; a.tiles_x = W / 16;
                a.total_tiles = B * a.tiles_y * a.tiles_x;
                a.wgs = r.sw.srd_wgs;
                char kn[64];
                of_first_kernel_name(prec, kn, sizeof kn);
                g_last_kernel = kn;
                const double px = (double)B * N * H * W;
                r.prof_begin(kn, p0, 2.0 * px * (9.0 * 3 * 8 + 9.0 * 8 * 8 + 3.0 * 8), px * (3 * 4.0 + 8 * r.elem_bytes()));
                r.check(launch_of_first(prec, a, r.s), "of_first");
                r.prof_end();
            }
        } else {
            Act in = r.act(B, N, H, W, 8);
            if (r.ok() && !r.dry) {
                char kn[56];
                snprintf(kn, sizeof kn, "dffw::from_ncdhw_pad_kernel<%d>", prec);
                r.prof_begin(kn, "flow.stack_in", 0.0, (double)B * N * H * W * (3 * 4.0 + 8 * r.elem_bytes()));
                r.check(launch_from_ncdhw_pad(prec, FS, in.p, B, 3, 8, N, H, W, r.s), "from_ncdhw_pad");
                r.prof_end();
            }
            a0 = of_block(r, p0, in);
            r.drop(in);
        }
    }
    Act fe1 = of_block(r, P + ".OF_feature.1", a0);
    r.drop(a0);
    Act a1 = of_block(r, P + ".OF_feature1.0", fe1);
    Act fe2 = of_block(r, P + ".OF_feature1.1", a1);
    r.drop(a1);
    Act a2 = of_block(r, P + ".OF_feature2.0", fe2);
    Act fe3 = of_block(r, P + ".OF_feature2.1", a2);
    r.drop(a2);

    const int64_t na = (int64_t)B * 3 * N;
    float *alpha = (float *)r.raw(na * sizeof(float));   // accumulated (scale offset, x shift, y shift) per (b, slice)
    float *rawh = (float *)r.raw(na * sizeof(float));    // last head output before damping (debug tap)
    if (r.ok() && !r.dry) r.check(hipMemsetAsync(alpha, 0, na * sizeof(float), r.s), "alpha memset");

    struct Level { Act *fe; const char *head; const char *tap; };
    Level levels[3] = {{&fe3, ".conv1", "head3"}, {&fe2, ".conv2", "head2"}, {&fe1, ".conv3", "head1"}};
    for (const Level &lv : levels) {                      // coarse to fine, End_to_End.py:77-103
        Act &fe = *lv.fe;
        const std::string hp = P + lv.head;
        char kn[56];
        snprintf(kn, sizeof kn, "dffw::flow_volume_kernel<%d>", prec);
        Act y0;
        if (r.e->convs.count(hp + ".0.0#ref") && !r.sw.on(SW_NO_HEAD_SPLIT)) {
            // the head's first conv is linear in its input channels: the part over the warped reference slice is the same
            // for all N slices of a sample, so it runs once per sample (1/N of the work, no ref channels in the volume) and
            // enters the per-slice conv over [cur | flow] as a slice-broadcast residual in front of the ReLU
            Act refw = r.act(B, 1, fe.H, fe.W, fe.C);
            // [cur | flow] is not materialised when head_warp_kernel serves the level (8- and 16-channel levels, whole 8 x 16 columns): it
            // samples the warped features while staging its tiles.  (The same inside conv_tile's fill was measured slower than
            // flow_volume + LDS-DMA fill -- 1.64 vs 0.85 + 0.93 ms at level 1: a tile's gathers are one dependent latency chain per
            // workgroup there -- and removed again.)
            auto ccur = r.e->convs.find(hp + ".0.0#cur");
            const bool roll = (fe.C == 8 || fe.C == 16) && ccur != r.e->convs.end() && ccur->second.wsrd && ccur->second.def.cin == fe.C + 2 &&
                              ccur->second.def.cout == 2 * fe.C && fe.H % 8 == 0 &&
                              fe.W % 16 == 0 && (int64_t)B * (fe.H / 8) * (fe.W / 16) >= r.sw.roll_min_units && (int64_t)B * N <= head_warp_max_planes() && !r.sw.on(SW_NO_HEAD_WARP) && !r.sw.on(SW_NO_TILE);
            Act vol;
            if (!roll) vol = r.act(B, N, fe.H, fe.W, fe.C + 8);
            if (r.ok() && !r.dry) {
                r.prof_begin(kn, std::string("flow") + lv.head + ".volume", 0.0,
                             ((double)(roll ? 0 : fe.pixels()) * (2.0 * fe.C + 8) + (double)refw.pixels() * 2.0 * fe.C) * r.elem_bytes());
                r.check(launch_flow_volume(prec, fe.p, refw.p, alpha, fov, B, N, fe.H, fe.W, fe.C, 2, r.s), "flow_volume ref");
                if (!roll) r.check(launch_flow_volume(prec, fe.p, vol.p, alpha, fov, B, N, fe.H, fe.W, fe.C, 1, r.s), "flow_volume cur");
                r.prof_end();
            }
            if (!roll) r.drop(fe);
            // per-slice conv: the B reference slices are presented as the B slices of ONE sample so that the 5-slice tiles
            // are filled (same memory either way)
            Act refw1 = refw;
            refw1.B = 1; refw1.N = B;
            Act refpart = r.conv(hp + ".0.0#ref", refw1);
            refpart.B = B; refpart.N = 1;
            r.drop(refw);
            if (roll) {
                y0 = r.act(B, N, fe.H, fe.W, 2 * fe.C);
                if (r.ok() && !r.dry) {
                    HeadWarpArgs ha;
                    memset(&ha, 0, sizeof ha);
                    ha.fe = fe.p; ha.ref = refpart.p; ha.out = y0.p;
                    ha.w = ccur->second.wsrd; ha.bias = ccur->second.bias;
                    ha.alpha = alpha; ha.fov = fov;
                    ha.B = B; ha.N = N; ha.H = fe.H; ha.W = fe.W;
                    ha.tiles_y = fe.H / 8; ha.tiles_x = fe.W / 16;
                    ha.total_tiles = B * ha.tiles_y * ha.tiles_x;
                    ha.wgs = r.sw.srd_wgs;
                    char knw[64];
                    head_warp_kernel_name(prec, fe.C, knw, sizeof knw);
                    g_last_kernel = knw;
                    const double px = (double)fe.pixels();
                    r.prof_begin(knw, hp + ".0.0#cur", 2.0 * px * 9.0 * (fe.C + 2) * 2 * fe.C, (px * 3 + (double)refpart.pixels() * 2) * fe.C * r.elem_bytes());
                    r.check(launch_head_warp(prec, fe.C, ha, r.s), "head_warp");
                    r.prof_end();
                }
                r.drop(fe);
            } else {
                ConvOpt oc = rl;
                oc.res0 = &refpart;
                oc.res_bcast = true;
                y0 = r.conv(hp + ".0.0#cur", vol, oc);
                r.drop(vol);
            }
            r.drop(refpart);
        } else {
            const int Cv = 2 * fe.C + 8;                  // 2C+2 channels of End_to_End.py:81-84, padded to a multiple of 8
            Act vol = r.act(B, N, fe.H, fe.W, Cv);
            if (r.ok() && !r.dry) {
                r.prof_begin(kn, std::string("flow") + lv.head + ".volume", 0.0, (double)fe.pixels() * (2.0 * fe.C + Cv) * r.elem_bytes());
                r.check(launch_flow_volume(prec, fe.p, vol.p, alpha, fov, B, N, fe.H, fe.W, fe.C, 0, r.s), "flow_volume");
                r.prof_end();
            }
            r.drop(fe);
            y0 = r.conv(hp + ".0.0", vol, rl);
            r.drop(vol);
        }
        Act y2;
        auto c6 = r.e->convs.find(hp + ".6");
        const bool tail_sums = c6 != r.e->convs.end() && c6->second.whead && !r.sw.on(SW_NO_HEAD_SUMS);
        bool tail_done = false;
        {
            // two 16 -> 16 per-slice convs in a row (level-1 head at full resolution): one streaming kernel, the intermediate in LDS
            auto c2 = r.e->convs.find(hp + ".2.0"), c4 = r.e->convs.find(hp + ".4.0");
            const auto end = r.e->convs.end();
            if (y0.C == 16 && c2 != end && c4 != end && c2->second.wsrd && c4->second.wsrd && c2->second.def.cout == 16 && c4->second.def.cout == 16 &&
                c2->second.cin_all == 16 && c4->second.cin_all == 16 && y0.H % 8 == 0 && y0.W % 16 == 0 &&
                (int64_t)y0.B * (y0.H / 8) * (y0.W / 16) >= r.sw.roll_min_units && !r.sw.on(SW_NO_FUSED_OF) && !r.sw.on(SW_NO_TILE)) {
                // ... and when the head's tail runs as plane sums, the pair's output is not stored either: the kernel leaves nine
                // 16-channel vectors per (slice, column) and head_tail_finish_tiles does the rest
                const bool sums = tail_sums && c6->second.def.cin == 16 && !r.sw.on(SW_NO_HEAD_SUMS_FUSED);
                const int tiles_y = y0.H / 8, tiles_x = y0.W / 16;
                float *tsum = nullptr;
                double *seg = nullptr;
                if (sums) {
                    tsum = (float *)r.raw((int64_t)B * N * tiles_y * tiles_x * 18 * 16 * sizeof(float));
                    seg = (double *)r.raw(head_tail_tiles_scratch_bytes(B, N, 16));
                }
                else y2 = r.act(y0.B, y0.N, y0.H, y0.W, 16);
                if (r.ok() && !r.dry) {
                    if (r.e->ensure_zero_page() != DFFW_OK) { r.err = DFFW_EHIP; return r.err; }
                    SrdArgs a;
                    memset(&a, 0, sizeof a);
                    a.x = y0.p; a.out = sums ? (uint16_t *)tsum : y2.p;
                    a.w0 = c2->second.wsrd; a.w2 = c4->second.wsrd;
                    a.b0 = c2->second.bias; a.b2 = c4->second.bias;
                    a.zero = r.e->zero_page;
                    a.B = y0.B; a.N = y0.N; a.H = y0.H; a.W = y0.W;
                    a.tiles_y = tiles_y; a.tiles_x = tiles_x;
                    a.total_tiles = y0.B * a.tiles_y * a.tiles_x;
                    a.wgs = r.sw.srd_wgs;
                    char kn2[64];
                    of_roll_kernel_name(prec, false, kn2, sizeof kn2, sums);
                    g_last_kernel = kn2;
                    const double px = (double)y0.pixels();
                    r.prof_begin(kn2, hp + (sums ? ".2.0+.4.0+.6+mean" : ".2.0+.4.0"), 2.0 * px * (2 * 9.0 * 16 * 16 + (sums ? 9.0 * 16 * 3 : 0.0)),
                                 px * (sums ? 16 : 32) * r.elem_bytes());
                    r.check(launch_of_roll(prec, false, a, r.s, sums), "of_roll (head)");
                    r.prof_end();
                    if (sums) {
                        r.prof_begin("dffw::head_tail_tiles_reduce_kernel", hp + ".6+mean (finish)", 0.0, (double)B * N * tiles_y * tiles_x * 18 * 16 * 4.0);
                        r.check(launch_head_tail_tiles(tsum, seg, tiles_y, tiles_x, c6->second.whead, alpha, rawh, B, N, y0.H, y0.W, 16, r.s), "head_tail_tiles");
                        r.prof_end();
                    }
                }
                r.drop(y0);
                if (sums) {
                    r.drop_raw(seg);
                    r.drop_raw(tsum);
                    tail_done = true;
                }
            } else {
                Act y1 = r.conv(hp + ".2.0", y0, rl);
                r.drop(y0);
                // the head's tail as plane sums and a row-sums kernel for the third conv: its output is not stored either
                if (tail_sums && c6->second.def.cin == y1.C && c4 != end && c4->second.def.cout == y1.C && !r.sw.on(SW_NO_HEAD_SUMS_FUSED) &&
                    r.sums_conv_ok(hp + ".4.0", y1.B, y1.N, y1.H, y1.W)) {
                    const int tiles_x = y1.W / c4->second.tile.cfg->tx;
                    float *rows = (float *)r.raw((int64_t)B * N * y1.H * tiles_x * 3 * y1.C * sizeof(float));
                    double *seg = (double *)r.raw(head_tail_tiles_scratch_bytes(B, N, y1.C));
                    ConvOpt os = rl;
                    os.sums = rows;
                    r.conv(hp + ".4.0", y1, os);
                    if (r.ok() && !r.dry) {
                        r.prof_begin("dffw::head_tail_rows_reduce_kernel", hp + ".6+mean (finish)", 0.0, (double)B * N * y1.H * tiles_x * 3 * y1.C * 4.0);
                        r.check(launch_head_tail_rows(rows, seg, tiles_x, c6->second.whead, alpha, rawh, B, N, y1.H, y1.W, y1.C, r.s), "head_tail_rows");
                        r.prof_end();
                    }
                    r.drop_raw(seg);
                    r.drop_raw(rows);
                    tail_done = true;
                } else {
                    y2 = r.conv(hp + ".4.0", y1, rl);
                }
                r.drop(y1);
            }
        }
        const int64_t hw = (int64_t)fe.H * fe.W;
        if (tail_done) {
        } else if (tail_sums && c6->second.def.cin == y2.C) {
            // last conv + plane mean collapsed into plane sums of y2 (dffw_kernels.hip, "alpha head tail"): y2 is read once, the
            // 3-plane fp32 head output is never formed
            const int nchunk = head_tail_chunks(B, N, hw);
            double *partial = (double *)r.raw((int64_t)B * N * (nchunk + 4) * y2.C * sizeof(double));
            if (r.ok() && !r.dry) {
                char kn6[64];
                snprintf(kn6, sizeof kn6, "dffw::plane_sums_kernel<%d>", prec);
                r.prof_begin(kn6, hp + ".6+mean", 2.0 * (double)y2.pixels() * 9.0 * y2.C * 3, (double)y2.pixels() * y2.C * r.elem_bytes());
                r.check(launch_head_tail(prec, y2.p, partial, c6->second.whead, alpha, rawh, B, N, y2.H, y2.W, y2.C, r.s), "head_tail");
                r.prof_end();
            }
            r.drop_raw(partial);
            r.drop(y2);
        } else {
            float *hf = (float *)r.raw(na * hw * sizeof(float));
            ConvOpt of; of.outf = hf; of.outf_ch = 3;
            r.conv(hp + ".6", y2, of);
            r.drop(y2);
            if (r.ok() && !r.dry) {
                r.prof_begin("dffw::alpha_mean_kernel", std::string("flow") + lv.head + ".mean", 0.0, (double)na * hw * 4.0);
                r.check(launch_alpha_mean(hf, alpha, rawh, B, N, hw, r.s), "alpha_mean");
                r.prof_end();
            }
            r.drop_raw(hf);
        }
        r.tap_f32(lv.tap, rawh, na);
    }
    r.tap_f32("alpha", alpha, na);
    if (r.ok() && !r.dry) {                               // End_to_End.py:104
        r.prof_begin("dffw::fov_warp_kernel", "flow.warp_stack", 0.0, (double)B * N * H * W * 3 * 8.0);
        r.check(launch_fov_warp(FS, alpha, fov, aligned, nullptr, B, 3, N, H, W, 0, r.s), "fov_warp");
        r.prof_end();
    }
    r.drop_raw(rawh);
    r.drop_raw(alpha);
    if (!r.ok()) return r.err;
    return run_depth(r, aligned, fd, fst, B, N, H, W, out);
}

static int check_dims(int B, int N, int H, int W) {
    if (B < 1 || N < 1) return fail(DFFW_EINVAL, "B and N must be >= 1 (got B=%d N=%d)", B, N);
    if (H < 32 || W < 32 || H % 32 || W % 32)
        return fail(DFFW_EINVAL, "H and W must be positive multiples of 32 (got %dx%d); pad with -1 as the reference loaders do", H, W);
    return DFFW_OK;
}

}  // namespace dffw

// ---- C ABI -------------------------------------------------------------------------------------
extern "C" {

const char *dffw_version(void) { return "dffw 0.1 (gfx950)"; }
const char *dffw_last_error(void) { return g_err.c_str(); }
const char *dffw_last_conv_kernel(void) { return g_last_kernel.c_str(); }

int dffw_param_count(int net) {
    if (!known_net(net)) return fail(DFFW_EINVAL, "unknown net %d", net);
    return (int)table_for(net).params.size();
}

int dffw_param_info(int net, int index, const char **name, int64_t shape[5], int *ndim, int *flags) {
    if (!known_net(net)) return fail(DFFW_EINVAL, "unknown net %d", net);
    const Table &t = table_for(net);
    if (index < 0 || index >= (int)t.params.size()) return fail(DFFW_EINVAL, "param index %d out of range", index);
    const ParamInfo &p = t.params[index];
    if (name) *name = p.name.c_str();
    if (shape) memcpy(shape, p.shape, sizeof p.shape);
    if (ndim) *ndim = p.ndim;
    if (flags) *flags = p.flags;
    return DFFW_OK;
}

int dffw_engine_create(int device, int net, const dffw_tensor *tensors, int n_tensors, int precision, dffw_engine **out) {
    if (!out) return fail(DFFW_EINVAL, "out is null");
    *out = nullptr;
    if (!known_net(net)) return fail(DFFW_EINVAL, "unknown net %d", net);
    if (precision < 0 || precision > 2) return fail(DFFW_EINVAL, "unknown precision %d", precision);
    if (!tensors || n_tensors <= 0) return fail(DFFW_EINVAL, "no tensors given");
    HIPCHK(hipSetDevice(device));
    std::map<std::string, const dffw_tensor *> by;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name) return fail(DFFW_EINVAL, "tensor %d has no name", i);
        std::string nm = tensors[i].name;
        if (nm.rfind("module.", 0) == 0) nm = nm.substr(7);
        by[nm] = &tensors[i];
    }
    auto need = [&](const std::string &nm, int64_t numel, const float **p) -> int {
        auto it = by.find(nm);
        if (it == by.end()) return fail(DFFW_EMISSING, "state dict has no entry '%s'", nm.c_str());
        if (it->second->numel != numel)
            return fail(DFFW_EINVAL, "'%s' has %lld elements, expected %lld", nm.c_str(), (long long)it->second->numel, (long long)numel);
        if (!it->second->data) return fail(DFFW_EINVAL, "'%s' has null data", nm.c_str());
        *p = it->second->data;
        return DFFW_OK;
    };
    std::unique_ptr<dffw_engine> e(new dffw_engine);
    e->device = device;
    e->net = net;
    e->prec = precision;
    const Table &t = table_for(net);
    for (const LayerDef &L : t.layers) {
        if (!L.live) continue;
        const float *w = nullptr, *cb = nullptr, *sw = nullptr;
        int scin = 0;
        if (!L.shortcut.empty()) {
            const LayerDef &S = t.layers[t.by_name.at(L.shortcut)];
            scin = S.cin;
            int rc2 = need(S.conv + ".weight", (int64_t)S.cin * S.cout, &sw);
            if (rc2) return rc2;
        }
        if (L.folded) continue;
        int rc = need(L.conv + ".weight", (int64_t)L.cin * L.cout * L.kd * L.kh * L.kw, &w);
        if (rc) return rc;
        if (L.bias && (rc = need(L.conv + ".bias", L.cout, &cb))) return rc;
        std::vector<float> bn;
        if (!L.bn.empty()) {
            const char *sfx[4] = {".weight", ".bias", ".running_mean", ".running_var"};
            bn.resize(4 * (size_t)L.cout);
            for (int k = 0; k < 4; ++k) {
                const float *p = nullptr;
                if ((rc = need(L.bn + sfx[k], L.cout, &p))) return rc;
                memcpy(bn.data() + (size_t)k * L.cout, p, L.cout * sizeof(float));
            }
        }
        PackedConv &pc = e->convs[L.conv];
        rc = pack_conv(L, precision, w, bn.empty() ? nullptr : bn.data(), cb, pc, sw, scin);
        if (rc) return rc;
        if (L.head_split > 0 && !bn.empty()) {
            // conv over [ref | cur | flow] = conv_ref(ref) + conv_cur([cur | flow]) (linearity): the ref part is the same for
            // every slice of a sample, so it is computed once per sample and added as a slice-broadcast residual
            const int C = L.head_split, kv = L.kd * L.kh * L.kw;
            auto slice = [&](int c0, int c1) {
                std::vector<float> ws((size_t)L.cout * (c1 - c0) * kv);
                for (int co = 0; co < L.cout; ++co)
                    memcpy(ws.data() + (size_t)co * (c1 - c0) * kv, w + ((size_t)co * L.cin + c0) * kv, (size_t)(c1 - c0) * kv * sizeof(float));
                return ws;
            };
            LayerDef Lr = L, Lc = L;
            Lr.conv = L.conv + "#ref"; Lr.cin = C; Lr.head_split = 0;
            Lc.conv = L.conv + "#cur"; Lc.cin = L.cin - C; Lc.head_split = 0;
            std::vector<float> bn_scale_only = bn;                       // gamma | beta | mean | var with beta = mean = 0: no shift
            for (int c = 0; c < L.cout; ++c) bn_scale_only[L.cout + c] = bn_scale_only[2 * L.cout + c] = 0.f;
            const std::vector<float> wr = slice(0, C), wc = slice(C, L.cin);
            if ((rc = pack_conv(Lr, precision, wr.data(), bn_scale_only.data(), nullptr, e->convs[Lr.conv]))) return rc;
            if ((rc = pack_conv(Lc, precision, wc.data(), bn.data(), nullptr, e->convs[Lc.conv]))) return rc;
        }
    }
    *out = e.release();
    return DFFW_OK;
}

void dffw_engine_destroy(dffw_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    delete e;
}

int dffw_engine_precision(const dffw_engine *e) { return e ? e->prec : fail(DFFW_EINVAL, "null engine"); }

int64_t dffw_workspace_bytes(const dffw_engine *e, int B, int N, int H, int W) {
    if (!e) return fail(DFFW_EINVAL, "null engine");
    int rc = check_dims(B, N, H, W);
    if (rc) return rc;
    if (e->net == DFFW_NET_E2E && N != DFFW_E2E_SLICES)
        return fail(DFFW_EINVAL, "the alignment network is built for %d focal slices (End_to_End.py:46), got %d", DFFW_E2E_SLICES, N);
    Run r(const_cast<dffw_engine *>(e), nullptr, true, nullptr, INT64_MAX / 2);
    const int64_t st[4] = {0, 0, 0, 0};
    float *outs[4] = {nullptr, nullptr, nullptr, nullptr};
    rc = e->net == DFFW_NET_E2E ? run_e2e(r, nullptr, nullptr, st, nullptr, B, N, H, W, outs, nullptr)
                                : run_depth(r, nullptr, nullptr, st, B, N, H, W, outs);
    if (rc) return rc;
    return r.arena.peak();
}

int dffw_forward_e2e(dffw_engine *e, const float *FS, const float *focus_dists, const int64_t fd_strides[4], const float *fovs,
                     int B, int N, int H, int W, float *const out[4], float *aligned, void *workspace, int64_t workspace_bytes,
                     void *hip_stream, const dffw_tap *taps, int n_taps) {
    if (!e || !FS || !focus_dists || !fd_strides || !fovs || !out || !aligned) return fail(DFFW_EINVAL, "null argument");
    if (e->net != DFFW_NET_E2E) return fail(DFFW_EINVAL, "engine was not created with DFFW_NET_E2E");
    int rc = check_dims(B, N, H, W);
    if (rc) return rc;
    if (N != DFFW_E2E_SLICES)
        return fail(DFFW_EINVAL, "the alignment network is built for %d focal slices (End_to_End.py:46), got %d", DFFW_E2E_SLICES, N);
    if (!workspace) return fail(DFFW_ENOMEM, "workspace is null");
    HIPCHK(hipSetDevice(e->device));
    if (e->profiling) e->clear_recs();
    Run r(e, (hipStream_t)hip_stream, false, (char *)workspace, workspace_bytes);
    r.taps = taps;
    r.n_taps = taps ? n_taps : 0;
    return run_e2e(r, FS, focus_dists, fd_strides, fovs, B, N, H, W, out, aligned);
}

int dffw_forward_taps(dffw_engine *e, const float *FS, const float *focus_dists, const int64_t fd_strides[4], int B, int N,
                      int H, int W, float *const out[4], void *workspace, int64_t workspace_bytes, void *hip_stream,
                      const dffw_tap *taps, int n_taps) {
    if (!e || !FS || !focus_dists || !fd_strides || !out) return fail(DFFW_EINVAL, "null argument");
    int rc = check_dims(B, N, H, W);
    if (rc) return rc;
    if (!workspace) return fail(DFFW_ENOMEM, "workspace is null");
    HIPCHK(hipSetDevice(e->device));
    if (e->profiling) e->clear_recs();
    Run r(e, (hipStream_t)hip_stream, false, (char *)workspace, workspace_bytes);
    r.taps = taps;
    r.n_taps = taps ? n_taps : 0;
    return run_depth(r, FS, focus_dists, fd_strides, B, N, H, W, out);
}

int dffw_forward_raw(dffw_engine *e, const void *raw, int dtype, const int64_t raw_strides[5], int h, int w, const float *focus_dists,
                     const int64_t fd_strides[4], int B, int N, int H, int W, float *const out[4], void *workspace,
                     int64_t workspace_bytes, void *hip_stream) {
    if (!e || !raw || !raw_strides || !focus_dists || !fd_strides || !out) return fail(DFFW_EINVAL, "null argument");
    if (e->net != DFFW_NET_DEPTH) return fail(DFFW_EINVAL, "dffw_forward_raw serves DFFW_NET_DEPTH engines");
    if ((dtype & ~DFFW_RAW_NORM_F64) != DFFW_RAW_U8 && (dtype & ~DFFW_RAW_NORM_F64) != DFFW_RAW_F32) return fail(DFFW_EINVAL, "unknown raw dtype %d", dtype);
    static_assert(DFFW_RAW_NORM_F64 == DFFW_RAW_NORM_F64_BIT, "dffw.h and dffw_internal.h disagree");
    int rc = check_dims(B, N, H, W);
    if (rc) return rc;
    if (h < 1 || w < 1 || h > H || w > W) return fail(DFFW_EINVAL, "source %dx%d does not fit the padded stack %dx%d", h, w, H, W);
    if (!workspace) return fail(DFFW_ENOMEM, "workspace is null");
    HIPCHK(hipSetDevice(e->device));
    if (e->profiling) e->clear_recs();
    Run r(e, (hipStream_t)hip_stream, false, (char *)workspace, workspace_bytes);
    RawStack rs{raw, dtype, raw_strides[0], raw_strides[1], raw_strides[2], raw_strides[3], raw_strides[4], h, w};
    return run_depth(r, nullptr, focus_dists, fd_strides, B, N, H, W, out, &rs);
}

int dffw_forward(dffw_engine *e, const float *FS, const float *focus_dists, const int64_t fd_strides[4], int B, int N, int H,
                 int W, float *const out[4], void *workspace, int64_t workspace_bytes, void *hip_stream) {
    return dffw_forward_taps(e, FS, focus_dists, fd_strides, B, N, H, W, out, workspace, workspace_bytes, hip_stream, nullptr, 0);
}

int dffw_profile_enable(dffw_engine *e, int on) {
    if (!e) return fail(DFFW_EINVAL, "null engine");
    e->profiling = on != 0;
    if (!on) e->clear_recs();
    return DFFW_OK;
}

int dffw_profile_collect(dffw_engine *e, dffw_prof_entry *out, int capacity) {
    if (!e) return fail(DFFW_EINVAL, "null engine");
    const int n = (int)e->recs.size();
    if (!out) return n;
    for (int i = 0; i < n && i < capacity; ++i) {
        ProfRec &r = e->recs[i];
        float ms = 0.f;
        if (r.e0 && r.e1) {
            HIPCHK(hipEventSynchronize(r.e1));
            HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        }
        out[i].kernel = r.kernel.c_str();
        out[i].layer = r.layer.c_str();
        out[i].flops = r.flops;
        out[i].bytes = r.bytes;
        out[i].ms = ms;
    }
    return n;
}

// ---- single-operator entry points --------------------------------------------------------------
int dffw_op_conv3d(int device, int precision, const float *x, int B, int Cin, int N, int H, int W, const float *weight,
                   int Cout, const int kernel[3], const int stride[3], const int pad[3], const int dilation[3], int transposed,
                   const float *bn, const float *conv_bias, const float *residual, int relu, float *y, void *hip_stream) {
    return dffw_op_conv3d_ex(device, precision, x, B, Cin, N, H, W, weight, Cout, kernel, stride, pad, dilation, transposed, bn, conv_bias, residual, relu, y,
                             nullptr, nullptr, nullptr, hip_stream);
}

int dffw_op_conv3d_ex(int device, int precision, const float *x, int B, int Cin, int N, int H, int W, const float *weight,
                      int Cout, const int kernel[3], const int stride[3], const int pad[3], const int dilation[3], int transposed,
                      const float *bn, const float *conv_bias, const float *residual, int relu, float *y, float *y_pre, const float *cls_weight,
                      float *cls_score, void *hip_stream) {
    if (!x || !weight || !y || !kernel || !stride || !pad || !dilation) return fail(DFFW_EINVAL, "null argument");
    if (precision < 0 || precision > 2) return fail(DFFW_EINVAL, "unknown precision %d", precision);
    if (stride[0] != 1 || dilation[0] != 1) return fail(DFFW_EINVAL, "slice stride/dilation must be 1");
    if (stride[1] != stride[2] || dilation[1] != dilation[2]) return fail(DFFW_EINVAL, "row/col stride and dilation must match");
    if (Cout != 1 && Cout % 4) return fail(DFFW_EINVAL, "Cout must be 1 or a multiple of 4");
    if (Cout > 128) return fail(DFFW_EINVAL, "Cout > 128 unsupported");
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    LayerDef L{"op", "", Cin, Cout, kernel[0], kernel[1], kernel[2], stride[1], stride[2], pad[0], pad[1], pad[2],
               dilation[1], dilation[2], transposed != 0, true, false};
    if (transposed && !(kernel[0] == 3 && kernel[1] == 3 && kernel[2] == 3 && stride[1] == 2 && pad[0] == 1 && pad[1] == 1 && pad[2] == 1))
        return fail(DFFW_EINVAL, "transposed conv supports only k3 s(1,2,2) p1 op(0,1,1)");
    dffw_engine eng;
    eng.device = device;
    eng.prec = precision;
    int rc = pack_conv(L, precision, weight, bn, conv_bias, eng.convs["op"]);
    if (rc) return rc;
    if ((cls_weight != nullptr) != (cls_score != nullptr)) return fail(DFFW_EINVAL, "cls_weight and cls_score go together");
    if ((y_pre || cls_weight) && (Cout == 1 || Cout % 8)) return fail(DFFW_EINVAL, "second output / fused classifier need Cout %% 8 == 0");
    if (cls_weight) {   // the 1x1x1 Cout -> 1 classifier applied to the final value (DEN.py:51-55), bias-free, no BatchNorm
        LayerDef C{"cls", "", Cout, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, false, true, false};
        rc = pack_conv(C, precision, cls_weight, nullptr, nullptr, eng.convs["cls"]);
        if (rc) return rc;
    }
    const int parts = prec_parts(precision);
    const int cpad = (Cin + 7) / 8 * 8;
    const bool stem = (!transposed && kernel[0] == 1 && kernel[1] == 9 && kernel[2] == 9 && dilation[1] == 2 && pad[0] == 0 && pad[1] == 8 &&
                       stride[1] == 1 && Cin == 3);
    Act in;
    in.B = B; in.N = N; in.H = H; in.W = stem ? W + 2 : W; in.C = cpad;
    const int64_t in_bytes = in.pixels() * parts * cpad * 2;
    HIPCHK(hipMalloc((void **)&in.p, in_bytes));
    HIPCHK(hipMemsetAsync(in.p, 0, in_bytes, s));
    if (stem) {
        HIPCHK(launch_stack_in(precision, x, in.p, B, N, H, W, s));   // the stem's paired-pixel input format
    } else if (cpad == Cin) {
        HIPCHK(launch_from_ncdhw(precision, x, in.p, B, Cin, N, H, W, s));
    } else {   // place the Cin real channels into the first channels of a zero-padded volume
        float *xp = nullptr;
        const int64_t plane = (int64_t)N * H * W;
        HIPCHK(hipMalloc((void **)&xp, (size_t)B * cpad * plane * sizeof(float)));
        HIPCHK(hipMemsetAsync(xp, 0, (size_t)B * cpad * plane * sizeof(float), s));
        for (int b = 0; b < B; ++b)
            HIPCHK(hipMemcpyAsync(xp + (int64_t)b * cpad * plane, x + (int64_t)b * Cin * plane, (size_t)Cin * plane * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
        HIPCHK(launch_from_ncdhw(precision, xp, in.p, B, cpad, N, H, W, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(xp));
    }
    int Ho, Wo;
    if (transposed) { Ho = 2 * H; Wo = 2 * W; }
    else {
        Ho = (H + 2 * pad[1] - dilation[1] * (kernel[1] - 1) - 1) / stride[1] + 1;
        Wo = (W + 2 * pad[2] - dilation[2] * (kernel[2] - 1) - 1) / stride[2] + 1;
    }
    const int No = N + 2 * pad[0] - (kernel[0] - 1);
    const int64_t opix = (int64_t)B * No * Ho * Wo;
    // run through the same Run::conv path the graph uses, on a private workspace
    const int64_t ws_bytes = 3 * (opix * parts * std::max(Cout, 4) * 2 + 4096) + 2 * (opix * 4 + 4096)
                             + 8 * opix * ((Cout + 15) / 16 * 16) * 4 + 4096;   // + split-K partial sums (up to 8 splits)
    char *ws = nullptr;
    HIPCHK(hipMalloc((void **)&ws, ws_bytes));
    Run r(&eng, s, false, ws, ws_bytes);
    Act res;
    ConvOpt o;
    o.relu = relu;
    float *scoref = nullptr;
    if (Cout == 1) {
        scoref = (float *)r.raw(opix * sizeof(float));
        o.outf = scoref;
    } else if (residual) {
        res = r.act(B, No, Ho, Wo, Cout);
        HIPCHK(launch_from_ncdhw(precision, residual, res.p, B, Cout, No, Ho, Wo, s));
        o.res0 = &res;
    }
    Act pre;
    float *clsf = nullptr;
    if (y_pre) o.out_pre = &pre;
    if (cls_weight) {
        clsf = (float *)r.raw(opix * sizeof(float));
        o.cls = "cls";
        o.cls_out = clsf;
    }
    Act out = r.conv("op", in, o);
    rc = r.err;
    if (rc == DFFW_OK) {
        if (Cout == 1) rc = hipMemcpyAsync(y, scoref, opix * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? DFFW_OK : fail(DFFW_EHIP, "copy");
        else rc = launch_to_ncdhw(precision, out.p, y, B, Cout, No, Ho, Wo, s) == hipSuccess ? DFFW_OK : fail(DFFW_EHIP, "to_ncdhw");
    }
    if (rc == DFFW_OK && y_pre) rc = launch_to_ncdhw(precision, pre.p, y_pre, B, Cout, No, Ho, Wo, s) == hipSuccess ? DFFW_OK : fail(DFFW_EHIP, "to_ncdhw");
    if (rc == DFFW_OK && cls_weight)
        rc = hipMemcpyAsync(cls_score, clsf, opix * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? DFFW_OK : fail(DFFW_EHIP, "copy");
    hipError_t se = hipStreamSynchronize(s);
    (void)hipFree(ws);
    (void)hipFree(in.p);
    if (rc == DFFW_OK && se != hipSuccess) rc = fail(DFFW_EHIP, "sync: %s", hipGetErrorString(se));
    return rc;
}

int dffw_op_pool(int device, int precision, int mode, int k, const float *x, int B, int C, int N, int H, int W, float *y,
                 void *hip_stream) {
    if (!x || !y) return fail(DFFW_EINVAL, "null argument");
    if (precision < 0 || precision > 2) return fail(DFFW_EINVAL, "unknown precision %d", precision);
    if (C % 8 || k < 1 || H % k || W % k) return fail(DFFW_EINVAL, "pool needs C %% 8 == 0 and H,W divisible by k");
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int parts = prec_parts(precision);
    uint16_t *a = nullptr, *b = nullptr;
    const int64_t nin = (int64_t)B * N * H * W * parts * C, nout = (int64_t)B * N * (H / k) * (W / k) * parts * C;
    HIPCHK(hipMalloc((void **)&a, nin * 2));
    HIPCHK(hipMalloc((void **)&b, nout * 2));
    int rc = DFFW_OK;
    hipError_t h = launch_from_ncdhw(precision, x, a, B, C, N, H, W, s);
    if (h == hipSuccess) h = launch_pool(precision, mode, k, a, b, B, N, H, W, C, s);
    if (h == hipSuccess) h = launch_to_ncdhw(precision, b, y, B, C, N, H / k, W / k, s);
    if (h == hipSuccess) h = hipStreamSynchronize(s);
    if (h != hipSuccess) rc = fail(DFFW_EHIP, "pool: %s", hipGetErrorString(h));
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

int dffw_op_regress(int device, const float *score, int B, int N, int h, int w, int H, int W, const float *focus_dists,
                    const int64_t fd_strides[4], float *depth, void *hip_stream) {
    if (!score || !focus_dists || !fd_strides || !depth) return fail(DFFW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(launch_regress(score, B, N, h, w, H, W, focus_dists, fd_strides[0], fd_strides[1], fd_strides[2], fd_strides[3], depth, s));
    HIPCHK(hipStreamSynchronize(s));
    return DFFW_OK;
}

int dffw_op_fov_warp(int device, const float *x, int B, int C, int N, int H, int W, const float *alpha, const float *fovs,
                     int alpha_from_sample0, float *out, float *flow, void *hip_stream) {
    if (!x || !alpha || !fovs || !out) return fail(DFFW_EINVAL, "null argument");
    if (B < 1 || C < 1 || N < 1 || H < 1 || W < 1) return fail(DFFW_EINVAL, "bad shape");
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(launch_fov_warp(x, alpha, fovs, out, flow, B, C, N, H, W, alpha_from_sample0, s));
    HIPCHK(hipStreamSynchronize(s));
    return DFFW_OK;
}

}  // extern "C"
