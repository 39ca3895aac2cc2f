// Host side of the register-blocked halo conv (kernel: conv3x_impl.h; instantiations: conv3x_<dtype>_<nd>.hip): the table of
// tilings, the default choice per layer shape and the launch entry points declared in kernels.h.
#include <vector>
#include "conv3x_impl.h"

namespace seg {
namespace {
using c3x::Conv3xArgs;

struct Cfg { int id, ndim, td, th, tw, bn, nres; const char* name; int cin16 = 0; };   // cin16: the Cin == 16 kernel (two taps per MFMA step)

// id, ndim, box, BN, resident chunks — kept in sync with SEG_C3X_3D_BODY / SEG_C3X_2D_BODY (conv3x_impl.h)
const Cfg kCfgs[] = {
    // 3-D
    {0, 3, 2, 8, 16, 32, 1, "2x8x16 t16 4x1 waves 4x2 tiles"},
    {1, 3, 4, 8, 16, 32, 1, "4x8x16 t16 4x1 waves 8x2 tiles"},
    {2, 3, 2, 8, 16, 64, 2, "2x8x16 t16 2x2 waves 8x2 tiles"},
    {3, 3, 2, 8, 8, 64, 2, "2x8x8 t8 2x2 waves 4x2 tiles"},
    {4, 3, 4, 8, 8, 64, 2, "4x8x8 t8 2x2 waves 8x2 tiles"},
    {5, 3, 4, 8, 8, 64, 2, "4x8x8 t8 4x1 waves 4x4 tiles"},
    {6, 3, 2, 8, 8, 64, 2, "2x8x8 t8 4x1 waves 2x4 tiles"},
    {7, 3, 2, 4, 12, 64, 4, "2x4x12 t4 2x2 waves 3x2 tiles"},
    {8, 3, 4, 4, 12, 64, 4, "4x4x12 t4 2x2 waves 6x2 tiles"},
    {9, 3, 2, 4, 12, 128, 4, "2x4x12 t4 2x2 waves 3x4 tiles"},
    {10, 3, 2, 8, 16, 16, 1, "2x8x16 t16 4x1 waves 4x1 tiles"},
    {11, 3, 2, 8, 8, 64, 4, "2x8x8 t8 2x2 waves 4x2 tiles, 4 resident chunks"},
    {12, 3, 2, 8, 8, 128, 4, "2x8x8 t8 2x2 waves 4x4 tiles, 4 resident chunks"},
    {13, 3, 2, 8, 8, 32, 2, "2x8x8 t8 4x1 waves 2x2 tiles"},
    {14, 3, 4, 8, 8, 32, 1, "4x8x8 t8 4x1 waves 4x2 tiles"},
    {15, 3, 2, 8, 16, 32, 2, "2x8x16 t16 4x1 waves 4x2 tiles, 2 resident chunks"},
    {16, 3, 2, 8, 16, 32, 1, "2x8x16 t16 4x1 waves 4x2 tiles, deep B ring, 2 workgroups/CU"},
    {17, 3, 4, 8, 8, 32, 1, "4x8x8 t8 4x1 waves 4x2 tiles, deep B ring, 2 workgroups/CU"},
    {20, 3, 4, 8, 8, 32, 1, "4x8x8 t8 4x1 waves 4x2 tiles, 3 workgroups/CU"},
    {21, 3, 2, 8, 8, 32, 1, "2x8x8 t8 4x1 waves 2x2 tiles, 1 resident chunk, 4 workgroups/CU"},
    {22, 3, 2, 8, 8, 32, 1, "2x8x8 t8 4x1 waves 2x2 tiles, 1 resident chunk, deep B ring, 3 workgroups/CU"},
    {23, 3, 4, 8, 8, 32, 1, "4x8x8 t8 4x1 waves 4x2 tiles, deep B ring, 3 workgroups/CU"},
    // the deep 12^3 / 6^3 levels are weight-stream latency chains (one L2 round trip per 8 steps of 6 MFMAs): a whole chunk of taps in flight
    // (PF = 26), and 32 output channels per workgroup = twice the workgroups, each streaming half the weights
    {44, 3, 2, 4, 12, 64, 4, "2x4x12 t4 2x2 waves 3x2 tiles, 26-deep B ring"},
    {45, 3, 2, 4, 12, 32, 4, "2x4x12 t4 2x2 waves 3x1 tiles, 26-deep B ring"},
    {46, 3, 2, 4, 12, 32, 4, "2x4x12 t4 2x2 waves 3x1 tiles"},
    {47, 3, 4, 4, 12, 32, 4, "4x4x12 t4 2x2 waves 6x1 tiles"},
    {48, 3, 4, 4, 12, 32, 4, "4x4x12 t4 2x2 waves 6x1 tiles, 26-deep B ring"},
    {49, 3, 2, 4, 12, 64, 4, "2x4x12 t4 1x4 waves 6x1 tiles (every wave its own weight columns)"},
    {50, 3, 2, 4, 12, 64, 4, "2x4x12 t4 1x4 waves 6x1 tiles, 26-deep B ring"},
    {24, 3, 2, 8, 16, 16, 1, "Cin16: 2x8x16 t16 4x1 waves 4x1 tiles", 1},
    {25, 3, 4, 8, 16, 16, 1, "Cin16: 4x8x16 t16 4x1 waves 8x1 tiles", 1},
    {26, 3, 2, 8, 16, 32, 1, "Cin16: 2x8x16 t16 4x1 waves 4x2 tiles", 1},
    {27, 3, 4, 8, 16, 32, 1, "Cin16: 4x8x16 t16 4x1 waves 8x2 tiles", 1},
    // Cin == 16, halo fragments reused across the kh taps (conv3x16r_kernel; weights packed with frag = 3: cin16 = 2)
    {28, 3, 2, 8, 16, 16, 1, "Cin16 row reuse: 2x8x16 t16 4x1 waves 4x1 tiles", 2},
    {29, 3, 4, 8, 16, 16, 1, "Cin16 row reuse: 4x8x16 t16 4x1 waves 8x1 tiles", 2},
    // 2-D
    {32, 2, 1, 16, 16, 32, 1, "16x16 t16 4x1 waves 4x2 tiles"},
    {33, 2, 1, 16, 16, 64, 2, "16x16 t16 2x2 waves 8x2 tiles"},
    {34, 2, 1, 8, 16, 64, 2, "8x16 t16 2x2 waves 4x2 tiles"},
    {35, 2, 1, 8, 16, 64, 4, "8x16 t16 2x2 waves 4x2 tiles, 4 resident chunks"},
    {36, 2, 1, 8, 16, 128, 4, "8x16 t16 2x2 waves 4x4 tiles, 4 resident chunks"},
    {37, 2, 1, 16, 16, 16, 1, "16x16 t16 4x1 waves 4x1 tiles"},
    {38, 2, 1, 8, 8, 64, 4, "8x8 t8 2x2 waves 2x2 tiles, 4 resident chunks"},
    {39, 2, 1, 8, 16, 32, 2, "8x16 t16 4x1 waves 2x2 tiles, 2 resident chunks"},
    {56, 2, 1, 16, 16, 16, 1, "Cin16: 16x16 t16 4x1 waves 4x1 tiles", 1},
    {57, 2, 1, 16, 16, 32, 1, "Cin16: 16x16 t16 4x1 waves 4x2 tiles", 1},
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

const Cfg* find_cfg(int id) {
    for (int i = 0; i < kNumCfgs; ++i)
        if (kCfgs[i].id == id) return &kCfgs[i];
    return nullptr;
}

bool cfg_fits(const Cfg& c, int ndim, int Cin, int Cout) {
    return c.ndim == ndim && Cout % c.bn == 0 && (c.cin16 != 0) == (Cin == 16);
}

}  // namespace

// ---- host entry points (declared in kernels.h) -----------------------------------------------------
bool conv3x_supported(int dtype, int ndim, int N, int D, int H, int W, int Cin, int Cout, int C0, bool has_in1) {
    (void)N;
    if (dtype == DT_F32) return false;
    if ((Cin % 32 && Cin != 16) || Cout % 16) return false;
    if (Cin == 16 && has_in1) return false;
    if (has_in1 && (C0 % 8 || C0 <= 0 || C0 >= Cin)) return false;
    const long long vol = (long long)(ndim == 3 ? D : 1) * H * W;
    if (vol * Cin * 2 >= (1ll << 31) || vol * 4 >= (1ll << 31)) return false;    // buffer ranges and the packed granule index stay below 2^31
    return true;
}

// default tiling per problem.  Override: SEG_C3X_MAP="cin:cout:w=id,..." (per layer shape; tools/tune_conv3x.py prints the measured table).
int conv3x_pick(int ndim, int N, int D, int H, int W, int Cin, int Cout, bool has_in1) {
    static const char* map = knob_s("SEG_C3X_MAP");
    if (map) {
        for (const char* p = map; *p;) {
            int ci = 0, co = 0, w = 0, id = -1;
            if (sscanf(p, "%d:%d:%d=%d", &ci, &co, &w, &id) == 4 && ci == Cin && co == Cout && w == W) {
                const Cfg* c = find_cfg(id);
                if (c && cfg_fits(*c, ndim, Cin, Cout)) return id;
            }
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
    }
    // measured choice per layer class (profiles/r02_tune_conv3x.jsonl: MI355X, every tiling on the halo-conv shapes of BASELINE
    // configs C2 - C5): 64 output channels per workgroup wherever Cout allows; 128-voxel boxes with all of Cin resident for the
    // wide levels, 96-voxel 4x4-tile boxes (3 x 2 register tiles) once the volume is small enough that workgroup count matters
    // more than operand reuse; for 32 output channels a 256-voxel box and 4 x 2 register tiles.
    const long long vox = (long long)N * D * H * W;
    int id;
    if (Cin == 16) {
        // 3-D: the row-reuse kernel (28 .. 31 = 24 .. 27 with a third of the LDS fragment reads); SEG_C3X16_REUSE=0: the one-read-per-MFMA kernel
        static const int reuse = knob_i("SEG_C3X16_REUSE", 1);
        // (measured and not kept, profiles/r06_conv3x16_row_reuse_*: the row-reuse form with two output tiles per wave = 32 output channels, 175 / 162 vs
        // 165 / 157 us at 2 x 128^3; four / five workgroups per CU instead of three / four: equal inside the step)
        id = ndim == 3 ? (Cout % 32 ? (vox >= (1ll << 21) ? 25 : 24) + (reuse ? 4 : 0) : 26) : (Cout % 32 ? 56 : 57);      // 25: 4x8x16 boxes, 67.6 vs 72.2 us at 4 x 96^3
        // (the persistent double-buffered tilings 28 / 29 lose: 106 us against 73.7 us at 4 x 96^3 standalone, same log)
    }
    else if (ndim == 3) {
        if (Cout % 32) id = 10;
        else if (Cout % 64) {
            // 32 -> 32: tiling 17 (deep B ring) since the epilogue stores from the accumulators: 912 vs 903 volumes/s against tiling 14 inside the
            // step, 33.4 vs 36.4 us standalone at 4 x 48^3 (profiles/r03_tiling_ab.log; round 2 had measured 14 ahead)
            id = Cin <= 32 ? 17 : (Cin == 64 ? (W >= 48 ? 17 : 15) : 14);
            // (tilings 18 / 19, persistent workgroups with LDS-resident weights and a double-buffered halo, are NOT the default: measured on
            // MI355X at 4 x 48^3 they run 37.4 us standalone against 35.0 / 37.8 us of tilings 17 / 14 and 70 against 61 us inside the train
            // step - one 144 KB workgroup per CU hides less than two 46 KB ones; profiles/r03_persistent_conv_ab.log)
            (void)has_in1;
        }
        else if (vox <= 16384) {
            id = Cout >= 2 * Cin ? 13 : 7;
            // the deepest level: a launch of 64-channel workgroups fills at most half the CUs and every workgroup is one serial weight stream from L2
            // (884 KB at 6^3 x 256 channels).  32 output channels per workgroup = twice the streams, each half as long, with a whole chunk of taps in
            // flight: 19.2 -> 14.4 us at 4 x 6^3 x 256, 19.0 -> 14.3 at 1 x 10^3, 19.4 -> 14.4 at 2 x 8^3, 11.7 -> 9.5 at 2 x 16^3 x 128 -> 64
            // (profiles/r04_deep_level_tilings.log); +1.0 ... +1.4 % on the train step on two leases.  Only while the doubled grid still fits the
            // 256 CUs in one round (12^3 x 128 x 4 samples: 144 -> 288 workgroups, 11.5 -> 16.3 us).
            const long long nb = (long long)N * ((D + 1) / 2) * ((H + 3) / 4) * ((W + 11) / 12);
            if (Cout % 64 == 0 && nb * (Cout / 64) <= 128) id = 45;
        }
        else id = Cin >= 128 ? 5 : 3;
    } else {
        if (Cout % 32) id = 37;
        else if (Cout % 64) id = 32;
        else id = Cin >= 256 ? 35 : 33;
    }
    const Cfg* c = find_cfg(id);
    if (c && cfg_fits(*c, ndim, Cin, Cout)) return id;
    for (int i = 0; i < kNumCfgs; ++i)
        if (cfg_fits(kCfgs[i], ndim, Cin, Cout)) return kCfgs[i].id;
    return -1;
}

// weight layout a tiling reads (seg_pack_desc.frag): 1 fragment-major per (chunk, tap), 2 flat k axis (Cin == 16), 3 taps paired within a kh row (conv3x16r_kernel)
int conv3x_cfg_frag(int cfg) {
    const Cfg* c = find_cfg(cfg);
    return !c ? 0 : (c->cin16 == 2 ? 3 : (c->cin16 ? 2 : 1));
}
int conv3x_num_cfgs() { return kNumCfgs; }
int conv3x_cfg_info(int index, int* id, int* ndim, int* box3, int* bn, int* nres, const char** name) {
    if (index < 0 || index >= kNumCfgs) return -1;
    const Cfg& c = kCfgs[index];
    if (id) *id = c.id;
    if (ndim) *ndim = c.ndim;
    if (box3) { box3[0] = c.td; box3[1] = c.th; box3[2] = c.tw; }
    if (bn) *bn = c.bn;
    if (nres) *nres = c.nres;
    if (name) *name = c.name;
    return 0;
}

bool launch_conv3x(int cfg, const void* in0, const void* in1, int C0, const void* w, const float* bias, void* out, double* stats, int N, int D,
                   int H, int W, int Cin, int Cout, int ndim, int dtype, hipStream_t s, int stat_rep) {
    const Cfg* c = find_cfg(cfg);
    if (!c || !cfg_fits(*c, ndim, Cin, Cout) || !conv3x_supported(dtype, ndim, N, D, H, W, Cin, Cout, C0, in1 != nullptr)) return false;
    Conv3xArgs a;
    a.in0 = in0; a.in1 = in1; a.C0 = in1 ? C0 : Cin; a.w = w; a.bias = bias; a.out = out; a.stats = stats;
    a.stat_rep = (stat_rep > 0 && stat_rep <= STAT_REP) ? stat_rep : STAT_REP;
    a.N = N; a.D = ndim == 3 ? D : 1; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
#ifdef SEG_C3X_TRACE
    // diagnostic build only (python tools/build_variant.py c3xtrace conv3x.hip,conv3x_f16_3d.hip -DSEG_C3X_TRACE; tools/trace_conv3x.py): per-workgroup phase stamps,
    // read back and summarised after every launch (synchronises the stream)
    static unsigned long long* tbuf = nullptr;
    const size_t tmax = 1 << 16;
    if (!tbuf) (void)hipMalloc(&tbuf, tmax * 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(tbuf, 0, tmax * 8 * sizeof(unsigned long long), s);
    a.trace = tbuf;
    struct TraceDump {
        unsigned long long* buf; size_t tmax; hipStream_t s; int cfg, N, D, H, W, Cin, Cout;
        ~TraceDump() {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h(tmax * 8);
            (void)hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            // phases between the stamps 0 -> 4 -> 5 -> 1 -> 2 -> 6 -> 7 -> 3 (a stamp that was not taken - no statistics - repeats its predecessor)
            const int order[8] = {0, 4, 5, 1, 2, 6, 7, 3};
            double ph[7] = {0, 0, 0, 0, 0, 0, 0}; size_t n = 0; unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t i = 0; i < tmax; ++i) {
                unsigned long long r[8];
                for (int k = 0; k < 8; ++k) r[k] = h[i * 8 + order[k]];
                if (!r[0] || !r[7]) continue;
                for (int k = 1; k < 8; ++k) if (!r[k]) r[k] = r[k - 1];
                ++n;
                for (int k = 0; k < 7; ++k) ph[k] += (double)(r[k + 1] - r[k]);
                t0 = r[0] < t0 ? r[0] : t0; t1 = r[7] > t1 ? r[7] : t1;
            }
            // wall_clock64 ticks at 100 MHz
            if (n) {
                double life = 0; for (int k = 0; k < 7; ++k) life += ph[k];
                fprintf(stderr, "[conv3x trace] cfg %d  %dx%dx%dx%d Cin %d Cout %d  workgroups %zu  first start -> last end %.2f us | mean per workgroup (us): issue copies %.2f  copies land %.2f  "
                                "barrier %.2f  taps %.2f  bias+convert+store %.2f  statistics butterfly %.2f  barrier+atomics %.2f  life %.2f\n",
                        cfg, N, D, H, W, Cin, Cout, n, (t1 - t0) * 0.01, ph[0] / n * 0.01, ph[1] / n * 0.01, ph[2] / n * 0.01, ph[3] / n * 0.01, ph[4] / n * 0.01, ph[5] / n * 0.01,
                        ph[6] / n * 0.01, life / n * 0.01);
            }
        }
    } dump{tbuf, tmax, s, cfg, N, a.D, H, W, Cin, Cout};
#endif
    a.remap = 1;                     // XCD-aware box order (c3x_box_of_block)
    if (ndim == 3) return dtype == DT_F16 ? c3x::launch_3d<f16>(cfg, a, s) : c3x::launch_3d<bf16>(cfg, a, s);
    return dtype == DT_F16 ? c3x::launch_2d<f16>(cfg, a, s) : c3x::launch_2d<bf16>(cfg, a, s);
}

}  // namespace seg
