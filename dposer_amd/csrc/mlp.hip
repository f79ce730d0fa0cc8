// TimeMLPs (reference lib/algorithms/advanced/model.py:69-90) on the GEMM family of the score network:
//     net = Linear(D + 1, H), act, [Linear(H, H), act, Dropout(p)] x n_blocks, Linear(H, D);    forward(x, t) = net(cat[x, t[:, None]])
// -- the reference's secondary model (run/train.py:163-170, `config.model.type == 'TimeMLPs'`).  Every Linear + activation is one
// gemm_ft_kernel launch with the EpiBiasSiLU epilogue (bias, activation, Philox dropout behind it in train mode, FT store; the training
// instantiation also keeps the pre-activation u), the last Linear one launch with EpiRowMajor; the backward pass is the dgrad GEMM with
// EpiSiLUBwd per hidden layer (dU = (dY W) * keep / (1 - p) * act'(u), column sums of dU = the bias gradient), the sample-major weight
// gradient kernel (bf16) or the plain one on transposed copies (fp32), and ONE deterministic reduction launch into the flat gradient.
// Same fragment-tiled layouts, tilings and packed-weight conventions as scorefc.hip; flat parameters in nn.Sequential order.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dposer_hip.h"
#include "gemm_api.h"
#include "kernels_api.h"

#define ML_HIP_LAUNCH(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return dposer_set_error(DPOSER_ERR_HIP, std::string(__func__) + ": " + #expr + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace {
constexpr int ML_MAX_LAYERS = 10;      // linear layers: n_blocks + 2
int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }
int64_t ml_pad_batch(int64_t B) { return B <= 512 ? round_up(B, 64) : round_up(B, 256); }
}   // namespace

struct dposer_mlp_s {
    dposer_mlp_desc d;
    int Din, Dout, H, Hp, NL;      // NL linear layers; hidden (activated) layers 0 .. NL-2; Hp = H padded to the 128-channel tiles (padded
                                   // channels carry zero weights and zero bias: act(0) = 0 for all four activations, so they stay exactly zero)
    int Kin, Cp;                   // padded input / output widths (multiples of 64)
    bool f32;                      // activations live in HBM as fp32 fragment tiles (precision fp32 AND bf16x3)
    bool w32;                      // exact-fp32 GEMMs (v_mfma_f32_32x32x2_f32): precision fp32 only
    bool x3;                       // bf16x3: GEMMs on the bf16 matrix pipe over hi / lo operand planes, weights packed [hi | lo | hi] per K segment (scorefc.hip)
    int esz, KBS;                  // bytes per stored activation element; k-block of the GEMM operands (bf16 and the bf16x3 planes: 16, fp32: 8)
    int wesz, wk;                  // bytes per packed weight element (2 / 4); K multiplier of the packed weights (3 in bf16x3 mode)
    int64_t w_off[ML_MAX_LAYERS], b_off[ML_MAX_LAYERS];      // flat offsets
    int kin[ML_MAX_LAYERS], kin_pad[ML_MAX_LAYERS], nout[ML_MAX_LAYERS], nout_pad[ML_MAX_LAYERS];
    int64_t pk_w[ML_MAX_LAYERS], pk_wT[ML_MAX_LAYERS], pk_b[ML_MAX_LAYERS], pk_end;     // pk_b: zero-padded bias copies (Hp != H only)
    int64_t nparams;
    std::vector<PackJob> jobs;
};

static int ml_main_shape(int64_t Spad, int channels, int act) {      // (the 256 x 256 tiling compiles swish in)
    if (act == DPOSER_ACT_SWISH && Spad % 256 == 0 && Spad >= 16384 && channels % 256 == 0) return SHAPE_BIG;
    if (Spad % 128 == 0 && channels % 128 == 0) return SHAPE_MID;
    return SHAPE_SMALL;
}
static int ml_final_shape(int64_t Spad) { return (Spad % 128 == 0 && Spad > 16384) ? SHAPE_FINAL : SHAPE_FINAL_S; }
static int ml_wgrad_shape(int n_rows_pad, int k_rows_pad, int64_t Spad) {
    if (n_rows_pad % 128 != 0) return SHAPE_FINAL;
    if (k_rows_pad % 128 != 0) return SHAPE_WIDE64;
    if (Spad >= 32768 && n_rows_pad % 256 == 0 && k_rows_pad % 256 == 0) return SHAPE_BIG;
    return SHAPE_MID;
}
static int ml_ksplit(int64_t tiles, int64_t stages, int slots) {
    int ks = 1;
    while (ks < 32 && tiles * ks < slots && stages % (ks * 2) == 0 && stages / (ks * 2) >= 4) ks *= 2;
    return ks;
}

extern "C" int dposer_mlp_create(const dposer_mlp_desc* desc, dposer_mlp_t* out) {
    DP_CHECK_ARG(desc && out, "null argument");
    DP_CHECK_ARG(desc->in_dim >= 1 && desc->in_dim <= 512 && desc->out_dim >= 1 && desc->out_dim <= 512, "in_dim / out_dim must be in 1..512");
    DP_CHECK_ARG(desc->hidden_dim >= 1 && desc->hidden_dim <= 4096, "hidden_dim must be in 1..4096");
    DP_CHECK_ARG(desc->n_blocks >= 0 && desc->n_blocks + 2 <= ML_MAX_LAYERS, "n_blocks must be 0..8");
    DP_CHECK_ARG(desc->precision == DPOSER_PREC_BF16 || desc->precision == DPOSER_PREC_FP32 || desc->precision == DPOSER_PREC_BF16X3, "bad precision");
    DP_CHECK_ARG(desc->activation >= DPOSER_ACT_SWISH && desc->activation <= DPOSER_ACT_LRELU, "bad activation");
    DP_CHECK_ARG(desc->dropout_p >= 0.f && desc->dropout_p < 1.f, "dropout_p must be in [0,1)");
    auto* h = new dposer_mlp_s();
    h->d = *desc;
    h->Din = desc->in_dim; h->Dout = desc->out_dim; h->H = desc->hidden_dim; h->Hp = (int)round_up(h->H, 128); h->NL = desc->n_blocks + 2;
    h->Kin = (int)round_up(h->Din, 64); h->Cp = (int)round_up(h->Dout, 64);
    h->w32 = desc->precision == DPOSER_PREC_FP32;
    h->x3 = desc->precision == DPOSER_PREC_BF16X3;
    h->f32 = h->w32 || h->x3;
    h->esz = h->f32 ? 4 : 2; h->KBS = h->w32 ? 8 : 16;
    h->wesz = h->w32 ? 4 : 2; h->wk = h->x3 ? 3 : 1;
    int64_t off = 0, p = 0;
    for (int i = 0; i < h->NL; ++i) {
        h->kin[i] = i == 0 ? h->Din : h->H; h->kin_pad[i] = i == 0 ? h->Kin : h->Hp;
        h->nout[i] = i == h->NL - 1 ? h->Dout : h->H; h->nout_pad[i] = i == h->NL - 1 ? h->Cp : h->Hp;
        h->w_off[i] = off; off += (int64_t)h->nout[i] * h->kin[i];
        h->b_off[i] = off; off += h->nout[i];
    }
    h->nparams = off;
    // (bf16x3: one logical copy = three column groups [hi | lo | hi] of the single K segment, as scorefc.hip packs its weights)
    auto job = [&](int64_t dst, int64_t src, int ktot, int rows_pad, int kpad, int rows_valid, int cols_valid, int ld, int trans) {
        for (int gcol = 0; gcol < h->wk; ++gcol) {
            PackJob j;
            j.dst_off = dst; j.src_off = src; j.ktot = h->wk * ktot; j.koff = gcol * kpad; j.rows_pad = rows_pad; j.kpad = kpad; j.rows_valid = rows_valid;
            j.cols_valid = cols_valid; j.ld = ld; j.trans = trans; j.f32 = h->w32 ? 1 : 0; j.split = h->x3 ? (gcol == 1 ? 2 : 1) : 0;
            h->jobs.push_back(j);
        }
    };
    for (int i = 0; i < h->NL; ++i) {       // forward copy [nout_pad][kin_pad] and, for dgrad, the transposed one [kin_pad][nout_pad]
        h->pk_w[i] = p; p = al256(p + (int64_t)h->nout_pad[i] * h->kin_pad[i] * h->wesz * h->wk);
        job(h->pk_w[i], h->w_off[i], h->kin_pad[i], h->nout_pad[i], h->kin_pad[i], h->nout[i], h->kin[i], h->kin[i], 0);
        h->pk_wT[i] = p; p = al256(p + (int64_t)h->kin_pad[i] * h->nout_pad[i] * h->wesz * h->wk);
        job(h->pk_wT[i], h->w_off[i], h->nout_pad[i], h->kin_pad[i], h->nout_pad[i], h->kin[i], h->nout[i], h->kin[i], 1);
    }
    for (int i = 0; i < h->NL - 1; ++i) { h->pk_b[i] = p; if (h->Hp != h->H) p = al256(p + (int64_t)h->Hp * 4); }
    h->pk_end = p;
    *out = h;
    return DPOSER_OK;
}
extern "C" void dposer_mlp_destroy(dposer_mlp_t h) { delete h; }
extern "C" int64_t dposer_mlp_num_params(dposer_mlp_t h) { return h ? h->nparams : -1; }
extern "C" int64_t dposer_mlp_packed_bytes(dposer_mlp_t h) { return h ? h->pk_end : -1; }

extern "C" int dposer_mlp_pack(dposer_mlp_t h, const float* flat, void* packed, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && flat && packed, "null argument");
    DP_CHECK_ARG(((uintptr_t)flat & 15) == 0 && ((uintptr_t)packed & 255) == 0, "flat_params must be 16-B aligned, packed 256-B aligned");
    for (size_t i0 = 0; i0 < h->jobs.size(); i0 += MAX_PACK_JOBS) {
        PackJobs js;
        js.n = (int)(h->jobs.size() - i0 < (size_t)MAX_PACK_JOBS ? h->jobs.size() - i0 : (size_t)MAX_PACK_JOBS);
        for (int i = 0; i < js.n; ++i) js.job[i] = h->jobs[i0 + i];
        ML_HIP_LAUNCH(launch_pack(js, flat, packed, (hipStream_t)stream));
    }
    if (h->Hp != h->H) {
        ML_HIP_LAUNCH(hipMemsetAsync((char*)packed + h->pk_b[0], 0, (size_t)(h->pk_end - h->pk_b[0]), (hipStream_t)stream));
        for (int i = 0; i < h->NL - 1; ++i)
            ML_HIP_LAUNCH(hipMemcpyAsync((char*)packed + h->pk_b[i], flat + h->b_off[i], (size_t)h->H * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return DPOSER_OK;
}

namespace {
struct MlPlanes { char *hi, *lo; };      // bf16x3: the two bf16 FT planes of an fp32 FT array (the GEMM operands)
struct MlWs {
    int64_t Bpad;
    char *xin, *xinT, *dres, *dresT;
    char *pre[ML_MAX_LAYERS], *hb[ML_MAX_LAYERS], *hT[ML_MAX_LAYERS], *dU[ML_MAX_LAYERS], *dUT[ML_MAX_LAYERS];
    MlPlanes p_xin, p_dres, p_hb[ML_MAX_LAYERS], p_dU[ML_MAX_LAYERS];
    float *fbase, *part[ML_MAX_LAYERS], *cs_last, *slabs;     // fbase: start of the fp32 partials region (reduction offsets are relative to it)
    int64_t slab_elems, total;
};
void ml_layout(const dposer_mlp_s* h, int64_t B, char* base, MlWs& w) {
    std::memset(&w, 0, sizeof(w));
    const int64_t Bpad = ml_pad_batch(B);
    w.Bpad = Bpad;
    const int esz = h->esz, H = h->Hp, NH = h->NL - 1;
    int64_t p = 0;
    auto take = [&](int64_t bytes) { char* r = base + p; p = al256(p + bytes); return r; };
    w.xin = take(Bpad * h->Kin * esz);
    w.dres = take(Bpad * h->Cp * esz);
    for (int i = 0; i < NH; ++i) {
        w.pre[i] = take(Bpad * H * esz);
        w.hb[i] = take(Bpad * H * esz);
        w.dU[i] = take(Bpad * H * esz);
    }
    if (h->x3) {
        auto planes = [&](int64_t elems) { MlPlanes q; q.hi = take(elems * 2); q.lo = take(elems * 2); return q; };
        w.p_xin = planes(Bpad * h->Kin); w.p_dres = planes(Bpad * h->Cp);
        for (int i = 0; i < NH; ++i) { w.p_hb[i] = planes(Bpad * H); w.p_dU[i] = planes(Bpad * H); }
    }
    if (h->w32) {       // transposed operand copies of the plain wgrad kernel (bf16 reads sample-major)
        w.xinT = take(Bpad * h->Kin * esz);
        w.dresT = take(Bpad * h->Cp * esz);
        for (int i = 0; i < NH; ++i) { w.hT[i] = take(Bpad * H * esz); w.dUT[i] = take(Bpad * H * esz); }
    }
    w.fbase = (float*)(base + p);
    for (int i = 0; i < NH; ++i) w.part[i] = (float*)take((Bpad / 32) * (int64_t)H * 4);
    w.cs_last = (float*)take(ceil_div(Bpad, 2048) * (int64_t)h->Cp * 4);
    const int64_t stages = Bpad / (h->KBS * 4);
    int64_t slab = 0;
    for (int i = 0; i < h->NL; ++i) {
        const int shape = ml_wgrad_shape(h->nout_pad[i], h->kin_pad[i], Bpad);
        const int64_t tiles = (int64_t)(h->nout_pad[i] / (shape_ct(shape) * 32)) * (h->kin_pad[i] / (shape_st(shape) * 32));
        slab += (int64_t)ml_ksplit(tiles, stages, shape == SHAPE_BIG ? 256 : 512) * h->nout[i] * h->kin[i] * h->wk;      // (bf16x3: one slab set per product term)
    }
    w.slabs = (float*)take(slab * 4);
    w.slab_elems = slab;
    w.total = p;
}
DropoutCfg ml_drop(const dposer_mlp_s* h, bool train, int layer, uint64_t seed, uint32_t step) {
    DropoutCfg d;
    std::memset(&d, 0, sizeof(d));
    if (train && h->d.dropout_p > 0.f && layer >= 1) {        // the Dropout modules sit behind the activations of the BLOCKS (layers 1 .. n_blocks)
        d.p = h->d.dropout_p;
        d.scale = 1.0f / (1.0f - d.p);
        d.thr = (uint32_t)((1.0 - (double)d.p) * 65536.0);
        d.site = (uint32_t)(layer - 1);
        d.offset = step;
        d.seed = seed;
        d.groups_x4 = h->Hp / 8;
    }
    return d;
}
GemmArgs ml_gemm(const void* W, int w_blocks, int n_cblk, int n_sblk, const void* src, int kblocks, double flops) {
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = w_blocks; g.n_cblk = n_cblk; g.n_sblk = n_sblk; g.ksplit = 1;
    g.src[0] = src; g.seg_kblocks[0] = kblocks; g.nseg = 1; g.ktot_blocks = kblocks; g.alg_flops = flops;
    return g;
}
// bf16x3: the activation operand as the three plane segments (hi, hi, lo) that meet the packed weight columns [hi | lo | hi]
GemmArgs ml_gemm_x3(const void* W, int n_cblk, int n_sblk, const MlPlanes& pl, int kblocks, double flops) {
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = 3 * kblocks; g.n_cblk = n_cblk; g.n_sblk = n_sblk; g.ksplit = 1;
    g.src[0] = pl.hi; g.src[1] = pl.hi; g.src[2] = pl.lo;
    for (int i = 0; i < 3; ++i) g.seg_kblocks[i] = kblocks;
    g.nseg = 3; g.ktot_blocks = 3 * kblocks; g.alg_flops = flops;
    return g;
}
}   // namespace

extern "C" int64_t dposer_mlp_workspace_bytes(dposer_mlp_t h, int64_t batch) {
    if (!h || batch <= 0) return -1;
    MlWs w;
    ml_layout(h, batch, nullptr, w);
    return w.total;
}

// out [B][out_dim] = net(x [B][in_dim]); train_mode != 0: dropout (Philox(seed, step, block)) and everything the backward needs is kept in ws
extern "C" int dposer_mlp_forward(dposer_mlp_t h, const float* flat, const void* packed_, void* ws_, const float* x, float* out, int64_t B,
                                  int32_t train_mode, int32_t keep_for_backward, uint64_t seed, uint32_t step, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && flat && packed_ && ws_ && x && out && B > 0, "bad argument");
    DP_CHECK_ARG(((uintptr_t)packed_ & 255) == 0 && ((uintptr_t)ws_ & 255) == 0, "packed / workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    MlWs w;
    ml_layout(h, B, (char*)ws_, w);
    const int prec = h->x3 ? PREC_BF16X3 : (h->w32 ? PREC_FP32 : PREC_BF16), KBS = h->KBS, NH = h->NL - 1;
    const bool keep = keep_for_backward != 0;
    ML_HIP_LAUNCH(launch_pack_rows(x, w.xin, B, w.Bpad, h->Din, h->Kin, h->f32, st));
    if (h->x3) ML_HIP_LAUNCH(launch_split_ft32(w.xin, w.p_xin.hi, w.p_xin.lo, w.Bpad, h->Kin, st));
    const void* in = w.xin;
    const MlPlanes* in_pl = &w.p_xin;
    for (int i = 0; i < NH; ++i) {
        const int shape = ml_main_shape(w.Bpad, h->Hp, h->d.activation);
        const int n_cblk = h->Hp / (shape_ct(shape) * 32), n_sblk = (int)(w.Bpad / (shape_st(shape) * 32));
        const double flops = 2.0 * (double)B * h->H * h->kin[i];
        GemmArgs g = h->x3 ? ml_gemm_x3(packed + h->pk_w[i], n_cblk, n_sblk, *in_pl, h->kin_pad[i] / KBS, flops)
                           : ml_gemm(packed + h->pk_w[i], h->kin_pad[i] / KBS, n_cblk, n_sblk, in, h->kin_pad[i] / KBS, flops);
        BiasSiLUParams p;
        std::memset(&p, 0, sizeof(p));
        p.bias = h->Hp != h->H ? (const float*)(packed + h->pk_b[i]) : flat + h->b_off[i]; p.out = w.hb[i]; p.pre = w.pre[i]; p.N = h->Hp;      // (the training instantiation always stores u)
         p.Spad = w.Bpad; p.act = h->d.activation;
        p.outT = (keep && h->w32) ? w.hT[i] : nullptr;
        p.out_hi = h->x3 ? w.p_hb[i].hi : nullptr;       // (bf16x3: the next layer's operand planes straight from the epilogue)
        p.out_lo = h->x3 ? w.p_hb[i].lo : nullptr;
        p.drop = ml_drop(h, train_mode != 0, i, seed, step);
        // (the inference instantiation has no dropout and keeps nothing; train mode or a kept graph take the training one)
        ML_HIP_LAUNCH(gemm_bias_silu(prec, keep || train_mode != 0, shape, g, p, st));
        in = w.hb[i];
        in_pl = &w.p_hb[i];
    }
    {
        const int shape = ml_final_shape(w.Bpad), i = h->NL - 1;
        const int n_cblk = h->Cp / (shape_ct(shape) * 32), n_sblk = (int)(w.Bpad / (shape_st(shape) * 32));
        const double flops = 2.0 * (double)B * h->Dout * h->H;
        GemmArgs g = h->x3 ? ml_gemm_x3(packed + h->pk_w[i], n_cblk, n_sblk, *in_pl, h->Hp / KBS, flops)
                           : ml_gemm(packed + h->pk_w[i], h->Hp / KBS, n_cblk, n_sblk, in, h->Hp / KBS, flops);
        RowMajorParams p;
        p.bias = flat + h->b_off[i]; p.out = out; p.ldc = h->Dout; p.C_valid = h->Dout; p.S_valid = B;
        ML_HIP_LAUNCH(gemm_rowmajor(prec, shape, g, p, st));
    }
    return DPOSER_OK;
}

// backward of dposer_mlp_forward(keep_for_backward = 1) on the same workspace: flat_grad (all parameters; may be NULL) and dx [B][in_dim] (may be NULL)
extern "C" int dposer_mlp_backward(dposer_mlp_t h, const float* flat, const void* packed_, void* ws_, const float* dout, float* flat_grad, float* dx,
                                   int64_t B, int32_t train_mode, uint64_t seed, uint32_t step, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && flat && packed_ && ws_ && dout && B > 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    MlWs w;
    ml_layout(h, B, (char*)ws_, w);
    const int prec = h->x3 ? PREC_BF16X3 : (h->w32 ? PREC_FP32 : PREC_BF16), KBS = h->KBS, NH = h->NL - 1, H = h->Hp;
    const int64_t Bpad = w.Bpad;
    const bool want_w = flat_grad != nullptr;
    ReduceJobs rj;
    rj.n = 0; rj.alt = nullptr;
    int64_t cursor = 0;
    auto rel = [&](const float* p) { return (int64_t)(p - w.fbase); };
    auto add_job = [&](int64_t dst, int64_t count, const float* src, int64_t stride, int nsrc) {
        ReduceJob& jb = rj.job[rj.n++];
        jb.dst_off = dst; jb.count = count; jb.src_off = rel(src); jb.src_stride = stride; jb.nsrc = nsrc;
    };
    // dW_i = dY_i^T in_i over the batch (sample-major kernel on bf16, transposed copies on fp32), split-K slabs -> one reduction job
    // (bf16x3: dW = dy_hi^T in_hi + dy_lo^T in_hi + dy_hi^T in_lo -- three sample-major launches on the planes into consecutive slab sets, one reduction job)
    auto wgrad = [&](int i, const void* dy, const void* dyT, const void* in, const void* inT, const MlPlanes* dy_pl = nullptr, const MlPlanes* in_pl = nullptr) -> int {
        const int shape = ml_wgrad_shape(h->nout_pad[i], h->kin_pad[i], Bpad);
        const int kb_total = (int)(Bpad / KBS);
        const int n_cblk = h->nout_pad[i] / (shape_ct(shape) * 32), n_sblk = h->kin_pad[i] / (shape_st(shape) * 32);
        const int ks = ml_ksplit((int64_t)n_cblk * n_sblk, kb_total / 4, shape == SHAPE_BIG ? 256 : 512);
        const int64_t numel = (int64_t)h->nout[i] * h->kin[i];
        WgradParams p;
        p.slab = w.slabs + cursor; p.slab_stride = numel; p.ld = h->kin[i]; p.N_valid = h->nout[i]; p.K_valid = h->kin[i];
        const double flops = 2.0 * (double)B * h->nout[i] * h->kin[i];
        int nterm = 1;
        if (h->x3) {
            nterm = 3;
            for (int term = 0; term < 3; ++term) {
                WgradTrArgs t;
                std::memset(&t, 0, sizeof(t));
                t.dY = term == 1 ? dy_pl->lo : dy_pl->hi; t.H = term == 2 ? in_pl->lo : in_pl->hi;
                t.N = h->nout_pad[i]; t.Kc = h->kin_pad[i]; t.n_cblk = n_cblk; t.n_sblk = n_sblk; t.sblocks = (int)(Bpad / 32); t.ksplit = ks;
                t.alg_flops = term == 0 ? flops : 0.0;
                WgradParams pt = p;
                pt.slab = p.slab + (int64_t)term * ks * numel;
                ML_HIP_LAUNCH(gemm_wgrad_tr(shape, t, pt, st));
            }
        } else if (!h->w32) {
            WgradTrArgs t;
            std::memset(&t, 0, sizeof(t));
            t.dY = dy; t.H = in; t.N = h->nout_pad[i]; t.Kc = h->kin_pad[i]; t.n_cblk = n_cblk; t.n_sblk = n_sblk; t.sblocks = (int)(Bpad / 32); t.ksplit = ks;
            t.alg_flops = flops;
            ML_HIP_LAUNCH(gemm_wgrad_tr(shape, t, p, st));
        } else {
            GemmArgs g = ml_gemm(dyT, kb_total, n_cblk, n_sblk, inT, kb_total, flops);
            g.ksplit = ks;
            ML_HIP_LAUNCH(gemm_wgrad(PREC_FP32, shape, g, p, st));
        }
        ReduceJob& j = rj.job[rj.n++];
        j.dst_off = h->w_off[i]; j.count = numel; j.src_off = rel(w.slabs + cursor); j.src_stride = numel; j.nsrc = ks * nterm;
        cursor += (int64_t)ks * nterm * numel;
        return DPOSER_OK;
    };
    // d res (FT, zero on padded rows / columns) and the last layer's parameter gradients
    ML_HIP_LAUNCH(launch_pack_rows(dout, w.dres, B, Bpad, h->Dout, h->Cp, h->f32, st));
    if (h->x3) ML_HIP_LAUNCH(launch_split_ft32(w.dres, w.p_dres.hi, w.p_dres.lo, Bpad, h->Cp, st));
    if (want_w) {
        int nch = 0;
        ML_HIP_LAUNCH(launch_colsum(h->f32, w.dres, w.cs_last, Bpad, h->Cp, &nch, st, nullptr));
        if (h->w32) {
            ML_HIP_LAUNCH(launch_ft_transpose(1, w.dres, w.dresT, Bpad, h->Cp, st));
            ML_HIP_LAUNCH(launch_ft_transpose(1, w.xin, w.xinT, Bpad, h->Kin, st));
        }
        DP_TRY(wgrad(h->NL - 1, w.dres, w.dresT, w.hb[NH - 1], h->w32 ? w.hT[NH - 1] : nullptr, &w.p_dres, &w.p_hb[NH - 1]));
        add_job(h->b_off[h->NL - 1], h->Dout, w.cs_last, h->Cp, nch);
    }
    for (int i = NH - 1; i >= 0; --i) {
        // dU_i = (dY_{i+1} W_{i+1}) * keep / (1 - p) * act'(u_i)
        const bool from_last = i == NH - 1;
        const int shape = ml_main_shape(Bpad, H, h->d.activation);
        const int kblocks = (from_last ? h->Cp : H) / KBS;
        const double dflops = 2.0 * (double)B * h->H * (from_last ? h->Dout : h->H);
        GemmArgs g = h->x3 ? ml_gemm_x3(packed + h->pk_wT[i + 1], H / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)), from_last ? w.p_dres : w.p_dU[i + 1], kblocks, dflops)
                           : ml_gemm(packed + h->pk_wT[i + 1], kblocks, H / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)),
                                     from_last ? (const void*)w.dres : (const void*)w.dU[i + 1], kblocks, dflops);
        SiLUBwdParams p;
        std::memset(&p, 0, sizeof(p));
        p.pre = w.pre[i]; p.out = w.dU[i]; p.N = H; p.S_valid = B; p.outT = (want_w && h->w32) ? w.dUT[i] : nullptr; p.Spad = Bpad; p.act = h->d.activation;
        p.part = want_w ? w.part[i] : nullptr;
        p.drop = ml_drop(h, train_mode != 0, i, seed, step);
        ML_HIP_LAUNCH(gemm_silu_bwd(prec, shape, g, p, st));
        if (h->x3) ML_HIP_LAUNCH(launch_split_ft32(w.dU[i], w.p_dU[i].hi, w.p_dU[i].lo, Bpad, H, st));      // dU is the next dgrad's and this layer's wgrad operand
        if (want_w) {
            DP_TRY(wgrad(i, w.dU[i], w.dUT[i], i == 0 ? (const void*)w.xin : (const void*)w.hb[i - 1], i == 0 ? (const void*)w.xinT : (const void*)w.hT[i - 1],
                         &w.p_dU[i], i == 0 ? &w.p_xin : &w.p_hb[i - 1]));
            const int rows = (int)(Bpad / (shape_st(shape) * 32)) * shape_ws(shape);
            add_job(h->b_off[i], h->H, w.part[i], H, rows);
        }
    }
    if (dx) {
        const int shape = ml_final_shape(Bpad);
        const double xflops = 2.0 * (double)B * h->H * h->Din;
        GemmArgs g = h->x3 ? ml_gemm_x3(packed + h->pk_wT[0], h->Kin / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)), w.p_dU[0], H / KBS, xflops)
                           : ml_gemm(packed + h->pk_wT[0], H / KBS, h->Kin / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)), w.dU[0], H / KBS, xflops);
        RowMajorParams p;
        p.bias = nullptr; p.out = dx; p.ldc = h->Din; p.C_valid = h->Din; p.S_valid = B;
        ML_HIP_LAUNCH(gemm_rowmajor(prec, shape, g, p, st));
    }
    if (want_w && rj.n > 0) ML_HIP_LAUNCH(launch_reduce_grads(rj, w.fbase, flat_grad, st));
    return DPOSER_OK;
}
