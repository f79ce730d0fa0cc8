// Host orchestration of the score network on MI355X + the C ABI of include/dposer_hip.h.
//
// A ScoreModelFC evaluation (reference lib/algorithms/advanced/model.py:141-196) is 2 + L GEMM
// launches with fused epilogues (L = 1 + 2*n_blocks GroupNorm layers):
//   per-sample t : emb -> [GEMM bias+SiLU] temb -> L x [GEMM K-concat(h, temb) + GN + SiLU (+residual)]
//                  -> [GEMM post_dense] -> elementwise tail
//   shared t     : the whole time branch collapses into a per-step bias row (table built once
//                  for all N steps with two fp32 GEMMs), the L layer GEMMs run on the x-path only.
// Training adds the mirrored dgrad chain (GroupNorm/SiLU/dropout backward fused into the dgrad
// epilogue), split-K wgrad GEMMs over transposed fragment-tiled copies, and a deterministic
// slab reduction into the flat gradient.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dposer_hip.h"
#include "gemm_api.h"
#include "kernels_api.h"

static thread_local std::string g_last_error;
// ---- roctx ranges (common.h: DP_RANGE) -------------------------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("DPOSER_ROCTX");
        if (!e || e[0] != '1') return;
        // (rocprofv3 --marker-trace records the ranges of rocprofiler-sdk's roctx library; roctracer's legacy libroctx64 is the fallback)
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!so) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(so, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(so, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void dposer_range_push(const char* name) { Roctx& r = roctx(); if (r.push) r.push(name); }
void dposer_range_pop() { Roctx& r = roctx(); if (r.pop) r.pop(); }

int dposer_set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
extern "C" const char* dposer_last_error(void) { return g_last_error.c_str(); }
extern "C" int dposer_abi_version(void) { return DPOSER_ABI_VERSION; }

#define DP_HIP_LAUNCH(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return dposer_set_error(DPOSER_ERR_HIP, std::string(__func__) + ": " + #expr + ": " + hipGetErrorString(_e)); \
    } while (0)

constexpr int MAX_L = 7;

struct LayerOff {
    int64_t w, b, wt, bt, gamma, beta;   // flat offsets (elements)
    int kin, kin_pad;
};

struct dposer_scorefc_s {
    dposer_scorefc_desc d;
    int D, Dpad, H, E, L, Cp;
    int gs;                    // channels per GroupNorm group = H / 32
    bool f32;                  // activations live in HBM as fp32 fragment tiles (precision fp32 AND bf16x3)
    bool x3;                   // DPOSER_PREC_BF16X3: GEMMs on the bf16 matrix pipe over hi / lo operand planes (three products per term), fp32 everywhere else
    bool w32;                  // packed weights are fp32 and the GEMMs run the exact-fp32 MFMA (precision fp32 only)
    int esz, KBS;              // bytes per stored activation element; k-block of the GEMM operands (8 fp32 / 16 bf16)
    int wk;                    // K multiplier of the packed weights: 3 in bf16x3 mode ([hi | lo | hi] per K segment), else 1
    std::vector<int64_t> toff, tnum;
    LayerOff layer[MAX_L];
    int64_t off_cond_w, off_cond_b, off_gauss, off_se_w, off_se_b, off_post_w, off_post_b;
    int64_t nparams;
    int64_t nograd_lo[2], nograd_hi[2];
    int n_nograd;
    // packed workspace (byte offsets)
    int64_t pk_wse, pk_wl[MAX_L], pk_wpost, pk_bias_cat, pk_wt_all32, pk_wse32, pk_fwd_end;
    int64_t pk_wpostT, pk_wlT[MAX_L], pk_wtT_all, pk_bwd_end;
    std::vector<PackJob> fwd_jobs, bwd_jobs;
    BiasCatJobs bias_jobs;
    const unsigned char* dbg_keep = nullptr;   // test hook: injected dropout keep decisions [L][dbg_keep_batch][H] (device), see drop_cfg
    int64_t dbg_keep_batch = 0;
    AdamPackArgs adam_pack;    // tensors / element ranges of the fused optimizer + re-pack step (filled by create(); n_tensors == 0: not available)
    std::vector<float> host_stage;   // staging for small H2D copies (labels)
    // second stream for the parameter-gradient side of the backward pass (small batches), see backward_core
    hipStream_t side = nullptr;
    hipEvent_t ev_start = nullptr, ev_layer[MAX_L] = {}, ev_time = nullptr, ev_join = nullptr;
};

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static PackJob mk_job(int64_t dst_off, int64_t src_off, int ktot, int koff, int rows_pad, int kpad, int rows_valid, int cols_valid,
                      int ld, int trans, int f32) {
    PackJob j;
    j.dst_off = dst_off; j.src_off = src_off; j.ktot = ktot; j.koff = koff; j.rows_pad = rows_pad; j.kpad = kpad;
    j.rows_valid = rows_valid; j.cols_valid = cols_valid; j.ld = ld; j.trans = trans; j.f32 = f32; j.split = 0;
    return j;
}

extern "C" int dposer_scorefc_create(const dposer_scorefc_desc* desc, dposer_scorefc_t* out) {
    DP_CHECK_ARG(desc && out, "null argument");
    DP_CHECK_ARG(desc->hidden_dim == 512 || desc->hidden_dim == 1024 || desc->hidden_dim == 2048,
                 "hidden_dim must be 512, 1024 or 2048: nn.GroupNorm(32, H) has H/32 channels per group and the fused MFMA "
                 "epilogues cover groups of 16, 32 (one accumulator tile per group, the shipped configuration) and 64 channels");
    DP_CHECK_ARG(desc->embed_dim > 0 && desc->embed_dim % 128 == 0, "embed_dim must be a multiple of 128");
    DP_CHECK_ARG(desc->n_blocks >= 1 && desc->n_blocks <= 3, "n_blocks must be 1..3");
    DP_CHECK_ARG(desc->data_dim > 0 && desc->data_dim <= 512, "data_dim must be in 1..512");
    DP_CHECK_ARG(desc->precision == DPOSER_PREC_BF16 || desc->precision == DPOSER_PREC_FP32 || desc->precision == DPOSER_PREC_BF16X3, "bad precision");
    DP_CHECK_ARG(desc->dropout_p >= 0.f && desc->dropout_p < 1.f, "dropout_p must be in [0,1)");
    DP_CHECK_ARG(desc->activation >= DPOSER_ACT_SWISH && desc->activation <= DPOSER_ACT_LRELU, "bad activation");
    DP_CHECK_ARG(desc->activation == DPOSER_ACT_SWISH || desc->hidden_dim == 1024,
                 "elu / relu / lrelu are built for hidden_dim 1024 (the tile-per-group epilogues); other widths are swish only");
    auto* h = new dposer_scorefc_s();
    h->d = *desc;
    h->D = desc->data_dim;
    h->Dpad = (int)round_up(h->D, 64);
    h->Cp = h->Dpad;
    h->H = desc->hidden_dim;
    h->gs = desc->hidden_dim / 32;
    h->E = desc->embed_dim;
    h->L = 1 + 2 * desc->n_blocks;
    h->x3 = desc->precision == DPOSER_PREC_BF16X3;
    h->w32 = desc->precision == DPOSER_PREC_FP32;
    h->f32 = h->w32 || h->x3;
    h->esz = h->f32 ? 4 : 2;
    h->KBS = h->w32 ? 8 : 16;
    h->wk = h->x3 ? 3 : 1;
    const int D = h->D, H = h->H, E = h->E, L = h->L;

    // ---- flat parameter layout = ScoreModelFC.parameters() order (model.py:98-139) ----------------
    int64_t off = 0;
    auto add = [&](int64_t n) { int64_t o = off; h->toff.push_back(o); h->tnum.push_back(n); off += n; return o; };
    h->layer[0].w = add((int64_t)H * D); h->layer[0].b = add(H);
    h->layer[0].wt = add((int64_t)H * E); h->layer[0].bt = add(H);
    h->off_cond_w = add((int64_t)H * H); h->off_cond_b = add(H);
    h->layer[0].gamma = add(H); h->layer[0].beta = add(H);
    h->layer[0].kin = D; h->layer[0].kin_pad = h->Dpad;
    h->off_gauss = -1;
    if (desc->embedding == DPOSER_EMB_FOURIER) h->off_gauss = add(E / 2);
    h->off_se_w = add((int64_t)E * E); h->off_se_b = add(E);
    for (int l = 1; l < L; ++l) {
        h->layer[l].w = add((int64_t)H * H); h->layer[l].b = add(H);
        h->layer[l].wt = add((int64_t)H * E); h->layer[l].bt = add(H);
        h->layer[l].gamma = add(H); h->layer[l].beta = add(H);
        h->layer[l].kin = H; h->layer[l].kin_pad = H;
    }
    h->off_post_w = add((int64_t)D * H); h->off_post_b = add(D);
    h->nparams = off;
    h->n_nograd = 1;
    h->nograd_lo[0] = h->off_cond_w; h->nograd_hi[0] = h->off_cond_b + H;       // pre_dense_cond: unused in forward (model.py:111)
    h->nograd_lo[1] = h->nograd_hi[1] = 0;
    if (h->off_gauss >= 0) { h->nograd_lo[1] = h->off_gauss; h->nograd_hi[1] = h->off_gauss + E / 2; h->n_nograd = 2; }

    // ---- packed workspace ----------------------------------------------------------------------------
    // bf16x3: every GEMM weight matrix is packed as THREE bf16 column groups per K segment, [hi | lo | hi] (hi = bf16(w), lo =
    // bf16(w - hi)), against activation segments (x_hi, x_hi, x_lo): w x ~ hi hi + lo hi + hi lo, fp32 accumulation (lo lo and the split
    // residuals are <= 2^-16 of a product).  The fp32 copies of the time branch (time-bias table GEMMs) exist in every mode.
    const int esz = h->w32 ? 4 : 2, f = h->w32 ? 1 : 0, wk = h->wk;
    int64_t p = 0;
    h->fwd_jobs.clear();
    h->bwd_jobs.clear();
    // one logical job = wk jobs in bf16x3 mode: column group g of segment [koff, koff + kpad) lands at wk * koff + g * kpad
    auto push = [&](std::vector<PackJob>& js, int64_t dst, int64_t src, int ktot, int koff, int rows_pad, int kpad, int rows_valid, int cols_valid,
                    int ld, int trans, int f32) {
        if (!h->x3 || f32) { js.push_back(mk_job(dst, src, ktot, koff, rows_pad, kpad, rows_valid, cols_valid, ld, trans, f32)); return; }
        for (int g = 0; g < 3; ++g) {
            PackJob j = mk_job(dst, src, 3 * ktot, 3 * koff + g * kpad, rows_pad, kpad, rows_valid, cols_valid, ld, trans, 0);
            j.split = g == 1 ? 2 : 1;
            js.push_back(j);
        }
    };
    h->pk_wse = p; p = align256(p + (int64_t)E * E * esz * wk);
    push(h->fwd_jobs, h->pk_wse, h->off_se_w, E, 0, E, E, E, E, E, 0, f);
    for (int l = 0; l < L; ++l) {
        const LayerOff& lo = h->layer[l];
        const int ktot = lo.kin_pad + E;
        h->pk_wl[l] = p; p = align256(p + (int64_t)H * ktot * esz * wk);
        push(h->fwd_jobs, h->pk_wl[l], lo.w, ktot, 0, H, lo.kin_pad, H, lo.kin, lo.kin, 0, f);
        push(h->fwd_jobs, h->pk_wl[l], lo.wt, ktot, lo.kin_pad, H, E, H, E, E, 0, f);
    }
    h->pk_wpost = p; p = align256(p + (int64_t)h->Cp * H * esz * wk);
    push(h->fwd_jobs, h->pk_wpost, h->off_post_w, H, 0, h->Cp, H, D, H, H, 0, f);
    h->pk_bias_cat = p; p = align256(p + (int64_t)L * H * 4);
    h->bias_jobs.n = L; h->bias_jobs.H = H;
    for (int l = 0; l < L; ++l) { h->bias_jobs.a_off[l] = h->layer[l].b; h->bias_jobs.b_off[l] = h->layer[l].bt; }
    h->pk_wt_all32 = p; p = align256(p + (int64_t)L * H * E * 4);
    for (int l = 0; l < L; ++l)
        push(h->fwd_jobs, h->pk_wt_all32 + (int64_t)l * H * E * 4, h->layer[l].wt, E, 0, H, E, H, E, E, 0, 1);
    if (h->w32) h->pk_wse32 = h->pk_wse;
    else {
        h->pk_wse32 = p; p = align256(p + (int64_t)E * E * 4);
        push(h->fwd_jobs, h->pk_wse32, h->off_se_w, E, 0, E, E, E, E, E, 0, 1);
    }
    h->pk_fwd_end = p;
    // backward: transposed copies (dgrad computes dX = dY @ W, i.e. "weights" = W^T)
    h->pk_wpostT = p; p = align256(p + (int64_t)H * h->Cp * esz * wk);
    push(h->bwd_jobs, h->pk_wpostT, h->off_post_w, h->Cp, 0, H, h->Cp, H, D, H, 1, f);
    for (int l = 0; l < L; ++l) {
        const LayerOff& lo = h->layer[l];
        h->pk_wlT[l] = p; p = align256(p + (int64_t)lo.kin_pad * H * esz * wk);
        push(h->bwd_jobs, h->pk_wlT[l], lo.w, H, 0, lo.kin_pad, H, lo.kin, H, lo.kin, 1, f);
    }
    if (h->x3) {
        // the time-branch dgrad runs one GEMM per layer in this mode (three activation segments each): one [E][3 H] matrix per layer
        h->pk_wtT_all = p; p = align256(p + (int64_t)E * L * H * esz * wk);
        for (int l = 0; l < L; ++l)
            push(h->bwd_jobs, h->pk_wtT_all + (int64_t)l * E * H * esz * wk, h->layer[l].wt, H, 0, E, H, E, H, E, 1, f);
    } else {
        h->pk_wtT_all = p; p = align256(p + (int64_t)E * L * H * esz);
        for (int l = 0; l < L; ++l)
            push(h->bwd_jobs, h->pk_wtT_all, h->layer[l].wt, L * H, l * H, E, H, E, H, E, 1, f);
    }
    h->pk_bwd_end = p;

    // ---- fused optimizer + re-pack step: every packed matrix with its copies, everything else as element ranges ------------------
    {
        AdamPackArgs& ap = h->adam_pack;
        std::memset(&ap, 0, sizeof(ap));
        int nt = 0, tiles = 0;
        auto none = [] { AdamPackDst d; d.off = -1; d.ktot = d.koff = d.trans = d.f32 = 0; return d; };
        auto dst = [&](int64_t off, int ktot, int koff, int trans, int f32) { AdamPackDst d; d.off = off; d.ktot = ktot; d.koff = koff; d.trans = trans; d.f32 = f32; return d; };
        auto add_t = [&](int64_t src, int R, int K, AdamPackDst d0, AdamPackDst d1, AdamPackDst d2) {
            AdamPackTensor& t = ap.tensor[nt++];
            t.src_off = src; t.R = R; t.K = K; t.ld = K; t.tile0 = tiles; t.dst[0] = d0; t.dst[1] = d1; t.dst[2] = d2;
            tiles += (int)(ceil_div(R, 64) * ceil_div(K, 64));
        };
        std::vector<std::pair<int64_t, int64_t>> matrices;        // [lo, hi) of the flat buffer covered by tensors
        auto mat = [&](int64_t off, int64_t n) { matrices.push_back({off, off + n}); };
        add_t(h->off_se_w, E, E, dst(h->pk_wse, E, 0, 0, f), h->w32 ? none() : dst(h->pk_wse32, E, 0, 0, 1), none());
        mat(h->off_se_w, (int64_t)E * E);
        for (int l = 0; l < L; ++l) {
            const LayerOff& lo = h->layer[l];
            const int ktot = lo.kin_pad + E;
            add_t(lo.w, H, lo.kin, dst(h->pk_wl[l], ktot, 0, 0, f), dst(h->pk_wlT[l], H, 0, 1, f), none());
            add_t(lo.wt, H, E, dst(h->pk_wl[l], ktot, lo.kin_pad, 0, f), dst(h->pk_wt_all32 + (int64_t)l * H * E * 4, E, 0, 0, 1),
                  dst(h->pk_wtT_all, L * H, l * H, 1, f));
            mat(lo.w, (int64_t)H * lo.kin);
            mat(lo.wt, (int64_t)H * E);
        }
        add_t(h->off_post_w, D, H, dst(h->pk_wpost, H, 0, 0, f), dst(h->pk_wpostT, h->Cp, 0, 1, f), none());
        mat(h->off_post_w, (int64_t)D * H);
        ap.n_tensors = nt;
        ap.n_tiles = tiles;
        // element ranges: the dense / dense_t bias pairs (their sums are the fp32 bias table), then every other tensor, neighbours merged
        int ne = 0, eblocks = 0;
        auto add_e = [&](int64_t a, int64_t b, int64_t cat, int64_t len) {
            AdamPackElems& e = ap.elems[ne++];
            e.off_a = a; e.off_b = b; e.cat_off = cat; e.len = (int)len; e.block0 = eblocks;
            int nb = (int)ceil_div(len, 1024);                    // one float4 per thread: the big dead range is ~1025 blocks of one trip each
            eblocks += nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
        };
        std::vector<std::pair<int64_t, int64_t>> taken = matrices;
        for (int l = 0; l < L; ++l) {
            add_e(h->layer[l].b, h->layer[l].bt, h->pk_bias_cat / 4 + (int64_t)l * H, H);
            taken.push_back({h->layer[l].b, h->layer[l].b + H});
            taken.push_back({h->layer[l].bt, h->layer[l].bt + H});
        }
        std::sort(taken.begin(), taken.end());
        int64_t cur = 0;
        bool ok = true;
        for (size_t i = 0; i <= taken.size(); ++i) {
            const int64_t lo_ = i < taken.size() ? taken[i].first : h->nparams;
            if (lo_ > cur) {
                if (ne >= ADAMPACK_MAX_ELEMS) { ok = false; break; }
                add_e(cur, -1, 0, lo_ - cur);
            }
            if (i < taken.size()) cur = taken[i].second > cur ? taken[i].second : cur;
        }
        ap.n_elems = ne;
        ap.n_elem_blocks = eblocks;
        ap.bias_cat_off = h->pk_bias_cat;
        if (!ok || nt > ADAMPACK_MAX_TENSORS || h->x3) ap.n_tensors = 0;      // (bf16x3: three column groups per segment -- the plain pack launch)
    }
    *out = h;
    return DPOSER_OK;
}

extern "C" void dposer_scorefc_destroy(dposer_scorefc_t h) {
    if (!h) return;
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipEventDestroy(h->ev_start);
        (void)hipEventDestroy(h->ev_time);
        (void)hipEventDestroy(h->ev_join);
        for (int l = 0; l < MAX_L; ++l) (void)hipEventDestroy(h->ev_layer[l]);
        (void)hipStreamDestroy(h->side);
    }
    delete h;
}
extern "C" int64_t dposer_scorefc_num_params(dposer_scorefc_t h) { return h ? h->nparams : -1; }
extern "C" int32_t dposer_scorefc_num_tensors(dposer_scorefc_t h) { return h ? (int32_t)h->toff.size() : -1; }
extern "C" int64_t dposer_scorefc_tensor_offset(dposer_scorefc_t h, int32_t i) { return (h && i >= 0 && i < (int)h->toff.size()) ? h->toff[i] : -1; }
extern "C" int64_t dposer_scorefc_tensor_numel(dposer_scorefc_t h, int32_t i) { return (h && i >= 0 && i < (int)h->tnum.size()) ? h->tnum[i] : -1; }
extern "C" int32_t dposer_scorefc_nograd_ranges(dposer_scorefc_t h, int64_t lo[2], int64_t hi[2]) {
    if (!h) return -1;
    for (int i = 0; i < 2; ++i) { lo[i] = h->nograd_lo[i]; hi[i] = h->nograd_hi[i]; }
    return h->n_nograd;
}
extern "C" int64_t dposer_scorefc_packed_bytes(dposer_scorefc_t h, int32_t with_backward) {
    if (!h) return -1;
    return with_backward ? h->pk_bwd_end : h->pk_fwd_end;
}

extern "C" int dposer_scorefc_pack(dposer_scorefc_t h, const float* flat, void* packed, int32_t with_backward, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && flat && packed, "null argument");
    DP_CHECK_ARG(((uintptr_t)flat & 15) == 0 && ((uintptr_t)packed & 255) == 0, "flat_params must be 16-B aligned, packed 256-B aligned");
    hipStream_t st = (hipStream_t)stream;
    // as few launches as the job table of one launch allows (the training step's tail is latency-bound): forward and backward operand
    // sets together where they fit (40 jobs; bf16 / fp32: 14 + 7 of them), in chunks otherwise (bf16x3: three jobs per matrix segment)
    std::vector<PackJob> all = h->fwd_jobs;
    if (with_backward) all.insert(all.end(), h->bwd_jobs.begin(), h->bwd_jobs.end());
    for (size_t i0 = 0; i0 < all.size(); i0 += MAX_PACK_JOBS) {
        PackJobs js;
        js.n = (int)(all.size() - i0 < (size_t)MAX_PACK_JOBS ? all.size() - i0 : (size_t)MAX_PACK_JOBS);
        for (int i = 0; i < js.n; ++i) js.job[i] = all[i0 + i];
        DP_HIP_LAUNCH(launch_pack(js, flat, packed, st));
    }
    DP_HIP_LAUNCH(launch_bias_cat(h->bias_jobs, flat, reinterpret_cast<float*>((char*)packed + h->pk_bias_cat), st));
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
// A/B switches of the score path: read from the environment ONCE (first use), not per call (three to four getenv() per backward
// pass in round 3); tests and tuners that change them inside one process call dposer_scorefc_tuning_reload() afterwards.
static int env_tri(const char* name) { const char* e = getenv(name); return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1; }
struct ScoreTuning {
    int64_t big_min = 16384;          // DPOSER_BIG_MIN_BATCH: 256x256 tiles from this many padded samples
    int gnbwd_big = -1;               // DPOSER_GNBWD_BIG = 0 / 1: force the 128x128 / 256x256 GroupNorm-backward dgrad
    int64_t final_small_max = 16384;  // DPOSER_FINAL_SMALL_MAX: 64x32 tiles for post_dense / dx up to this many samples
    int wgrad_big = -1;               // DPOSER_WGRAD_BIG = 0 / 1: force the 128x128 / 256x256 split-K wgrads
    int wgrad_tr = -1;                // DPOSER_WGRAD_TR = 0: bf16 wgrads on transposed operand copies
    int wgrad_stream = -1;            // DPOSER_WGRAD_STREAM = 0 / 1: parameter-gradient half of the backward on the second stream
    int wgrad_batched = -1;           // DPOSER_WGRAD_BATCHED = 0 / 1: never / always one lane launch for all 256x256 wgrad tiles
    int wgrad_layer_lanes = -1;       // DPOSER_WGRAD_LAYER_LANES = 0 / 1: bucketed backward with split-K launches / one lane launch per layer
    int wgrad_groups = -1;            // DPOSER_WGRAD_GROUPS = n: bucketed backward with the lane launches of n layer groups (0: off)
    int adam_write_through = 1;       // DPOSER_ADAM_WT = 0: plain stores for the optimizer state in the fused optimizer + re-pack kernel (A/B)
    int64_t silu_split_max = 2048;    // DPOSER_SILU_SPLIT_MAX = <samples>: up to this padded batch the time-branch dgrad runs one k-split per layer + a reduce pass (0: never)
    int dsm_fused = 0;                // DPOSER_DSM_FUSED = 1: post_dense with the DSM loss in its epilogue (EpiDsm) instead of GEMM -> res -> k_dsm
                                      // (opt-in: measured -0.6 % at 8192 poses, -0.2 % at 65536, +0.5 % at 1280 -- profiles/r04_dsm_fused_ab.txt)
    bool small64 = true;              // DPOSER_SMALL64=0: never the 128x64 / 8-wave tiling (A/B; bit-identical)
    int64_t small64_min = 1024, small64_max = 2048;      // DPOSER_SMALL64_MIN / _MAX: its window, min < padded samples <= max
    int64_t small_tile_max = 2048;    // DPOSER_SMALL_TILE_MAX = <samples>: up to this padded batch the GroupNorm layers take the 128x32 tiling also where 128x128
                                      // divides the batch (1280 samples are 80 workgroups of 128x128 on 256 CUs).  Bit-identical.  Round 6, with three K-loop slots on
                                      // that tiling (gemm_launch.hip): training step 0.370 -> 0.339 ms at 1536 poses, 0.389 -> 0.383 at 2048, slower from 2560
                                      // (profiles/r06_small_tile_max_sweep.md; round 5, two slots: the crossover was 1280, profiles/r05_small_tile_ab.txt)
    int sampler_persistent = 0;       // DPOSER_SAMPLER_PERSISTENT = 1: one persistent kernel for the plain EM sampler
    int64_t sampler_persistent_min = 256;
    void load() {
        const char* e = getenv("DPOSER_BIG_MIN_BATCH");
        big_min = e ? atoll(e) : (int64_t)16384;
        gnbwd_big = env_tri("DPOSER_GNBWD_BIG");
        e = getenv("DPOSER_FINAL_SMALL_MAX");
        final_small_max = e ? atoll(e) : (int64_t)16384;
        wgrad_big = env_tri("DPOSER_WGRAD_BIG");
        wgrad_tr = env_tri("DPOSER_WGRAD_TR");
        wgrad_stream = env_tri("DPOSER_WGRAD_STREAM");
        wgrad_batched = env_tri("DPOSER_WGRAD_BATCHED");
        wgrad_layer_lanes = env_tri("DPOSER_WGRAD_LAYER_LANES");
        e = getenv("DPOSER_WGRAD_GROUPS");
        wgrad_groups = e ? atoi(e) : -1;
        adam_write_through = env_tri("DPOSER_ADAM_WT") == 0 ? 0 : 1;
        dsm_fused = env_tri("DPOSER_DSM_FUSED") == 1 ? 1 : 0;
        e = getenv("DPOSER_SILU_SPLIT_MAX");
        silu_split_max = e ? atoll(e) : (int64_t)2048;
        e = getenv("DPOSER_SMALL64");
        small64 = !(e && e[0] == '0');
        e = getenv("DPOSER_SMALL64_MIN");
        small64_min = e ? atoll(e) : (int64_t)1024;
        e = getenv("DPOSER_SMALL64_MAX");
        small64_max = e ? atoll(e) : (int64_t)2048;
        e = getenv("DPOSER_SMALL_TILE_MAX");
        small_tile_max = e ? atoll(e) : (int64_t)2048;
        e = getenv("DPOSER_SAMPLER_PERSISTENT");
        sampler_persistent = e ? atoi(e) : 0;
        e = getenv("DPOSER_SAMPLER_PERSISTENT_MIN");
        sampler_persistent_min = e ? atoll(e) : (int64_t)256;
    }
};
static ScoreTuning& score_tuning() {
    static ScoreTuning t = [] { ScoreTuning x; x.load(); return x; }();
    return t;
}
extern "C" void dposer_scorefc_tuning_reload(void) { score_tuning().load(); }

static int64_t pad_batch(int64_t B) { return B <= 512 ? round_up(B, 64) : round_up(B, 256); }
// `channels` = output channels of the GEMM (H = 1024 for the GroupNorm layers, E for the time branch): the 256x256 tiling
// needs them to be a multiple of 256 (embed_dim may be any multiple of 128).
static thread_local int g_act = DPOSER_ACT_SWISH;   // activation of the handle whose call is running (the 256 x 256 tiling compiles swish in)
static int main_shape(int64_t Spad, int channels = 1024, int gs = 32) {
    if (Spad <= score_tuning().small_tile_max) return SHAPE_SMALL;
    if (g_act != DPOSER_ACT_SWISH) return Spad % 128 == 0 ? SHAPE_MID : SHAPE_SMALL;
    const int64_t big_min = score_tuning().big_min;
    if (gs == 32 && Spad % 256 == 0 && Spad >= big_min && channels % 256 == 0) return SHAPE_BIG;   // (generic group sizes: 128-wide tilings)
    if (Spad % 128 == 0) return SHAPE_MID;
    return SHAPE_SMALL;
}
// GroupNorm-backward dgrad: the register-lean epilogue fits the 256x256 tile in 248 VGPRs without spilling; it wins from
// 16384 samples up (227 vs 257 us at 65536, 2.52 vs 2.59 ms per step at 32768; round 5, re-measured: 0.960 vs 0.976 ms per step at 16384,
// 0.665 vs 0.621 at 8192 -- profiles/r05_small_tile_ab.txt).  DPOSER_GNBWD_BIG = 0 / 1 forces it.
// GroupNorm layers (forward and dgrad), bf16 storage, SiLU, 32-channel groups: 128 x 64 on eight waves between 1024 and 2048 samples (SHAPE_SMALL64)
static bool small64_window(int64_t Spad, int gs, bool bf16) {
    const ScoreTuning& tn = score_tuning();
    return bf16 && tn.small64 && gs == 32 && g_act == DPOSER_ACT_SWISH && Spad % 64 == 0 && Spad > tn.small64_min && Spad <= tn.small64_max;
}
static int gn_shape(int64_t Spad, int channels, int gs, bool bf16) { return small64_window(Spad, gs, bf16) ? SHAPE_SMALL64 : main_shape(Spad, channels, gs); }
static int gnbwd_shape(int64_t Spad, int gs = 32, bool bf16 = false) {
    if (small64_window(Spad, gs, bf16)) return SHAPE_SMALL64;
    if (Spad <= score_tuning().small_tile_max) return SHAPE_SMALL;
    const int forced = score_tuning().gnbwd_big;
    const bool big = gs == 32 && g_act == DPOSER_ACT_SWISH && (forced >= 0 ? forced == 1 : Spad >= 16384);
    if (big && Spad % 256 == 0) return SHAPE_BIG;
    return Spad % 128 == 0 ? SHAPE_MID : SHAPE_SMALL;
}
// post_dense / dx: one 64-channel block tile, so the grid is Spad/128 (or Spad/32) workgroups: below 32768 samples the
// 64x128 tiling leaves most CUs idle and the 64x32 one is used.
static int final_shape(int64_t Spad) {
    const int64_t small_max = score_tuning().final_small_max;
    return (Spad % 128 == 0 && Spad > small_max) ? SHAPE_FINAL : SHAPE_FINAL_S;
}

struct Planes { char *hi, *lo; };    // bf16x3 mode: the two bf16 FT planes of an fp32 FT activation (k_split_ft32), operands of the GEMMs
struct Ws {
    int64_t Bpad;
    Planes p_xin, p_emb, p_temb, p_h[MAX_L], p_dres, p_dy[MAX_L], p_dU;
    char *xin, *emb, *temb, *upre, *hbuf[MAX_L], *xhat[MAX_L], *dy[MAX_L], *carry[2], *dU, *dres;
    GnAux* aux[MAX_L];        // per GroupNorm layer: rstd + dropout decisions for the backward epilogue (epilogues.h)
    float *res, *xt, *tbuf, *zbuf, *loss_part, *scalar;
    float *xft, *xmft;        // sampler fast path: state and last x_mean as fp32 FT [Bpad][Dpad]
    // shared-t time table
    int64_t npad;
    float *tt_labels, *tt_emb, *tt_temb, *table;
    float* tt_t;                   // t of every step (persistent sampler)
    SamplerLayer* smp_layers;      // device table of the persistent sampler's per-layer operands
    uint32_t* smp_sync;            // cluster sampler: control words + one progress counter per block of 256 samples
    // transposed copies / partials / slabs (training)
    char *dyT[MAX_L], *hT[MAX_L], *tembT, *embT, *xinT, *dresT, *dUT;
    float *gn_part[MAX_L], *cs_part_post, *cs_part_se, *silu_part, *slabs, *dU_part;
    int64_t total;
    int64_t slab_elems;   // capacity of `slabs` in floats
};

struct WgradPlan {
    int shape, ksplit;
    int64_t slab_off;   // element offset inside Ws::slabs
};
// Tiling of a split-K wgrad GEMM over Spad samples.  256x256 tiles (one workgroup per CU, k-split sized for 256 resident
// workgroups) win from 32768 samples up (93 vs 109 us per 1024x1024 wgrad at 65536; measured slower at 16384 and below, where
// the per-split k-range gets too short).  DPOSER_WGRAD_BIG = 0 / 1 forces the choice.
static int wgrad_shape(int n_rows_pad, int k_rows_pad, int64_t Spad) {
    // (rows / columns come in multiples of 64: data dimensions of 129..192, e.g. 52 joints x 3, pad to 192 -- only the 64-row /
    //  64-column tilings divide that)
    if (n_rows_pad % 128 != 0) return SHAPE_FINAL;
    if (k_rows_pad % 128 != 0) return SHAPE_WIDE64;
    const int forced = score_tuning().wgrad_big;
    const bool big = forced >= 0 ? forced == 1 : Spad >= 32768;
    if (big && n_rows_pad % 256 == 0 && k_rows_pad % 256 == 0) return SHAPE_BIG;
    return SHAPE_MID;
}
// bf16 wgrads read their operands sample-major (gemm_wgrad_tr.h): the training epilogues then skip the transposed activation
// copies those GEMMs used to need (fp32 = parity mode keeps them: there is no 32-bit transposing LDS read).
// DPOSER_WGRAD_TR = 0 forces the transposed-copy path.
static bool wgrad_tr_mode(const dposer_scorefc_s* h, int64_t Bpad) {
    const int forced = score_tuning().wgrad_tr;
    (void)Bpad;
    return h->x3 || (forced != 0 && !h->f32);      // every wgrad tiling has a sample-major instantiation (bf16x3: on the bf16 planes, always)
}
static int pick_ksplit(int64_t tiles, int64_t stages, int slots = 512) {
    int ks = 1;
    while (ks < 32 && tiles * ks < slots && stages % (ks * 2) == 0 && stages / (ks * 2) >= 4) ks *= 2;
    return ks;
}

constexpr int64_t SILU_SPLIT_CAP = 4096;     // largest padded batch the k-split partial buffer of the time-branch dgrad is laid out for
static inline int64_t silu_part_rows(int64_t Bpad, bool always_reduce = false) { return (Bpad <= SILU_SPLIT_CAP || always_reduce) ? Bpad / 2 : Bpad / 32; }   // (the reduce pass: one block per two samples)
static inline int64_t cs_post_rows(int64_t Bpad) { return Bpad / 32 > 1024 ? Bpad / 32 : 1024; }   // capacity of Ws::cs_part_post in rows
static void layout_ws(const dposer_scorefc_s* h, int64_t B, int mode, int n_steps, char* base, Ws& w) {
    std::memset(&w, 0, sizeof(w));
    const int64_t Bpad = pad_batch(B);
    w.Bpad = Bpad;
    const int esz = h->esz, H = h->H, E = h->E, L = h->L;
    int64_t p = 0;
    auto take = [&](int64_t bytes) { char* r = base + p; p = align256(p + bytes); return r; };
    auto planes = [&](int64_t elems) { Planes q; q.hi = q.lo = nullptr; if (h->x3) { q.hi = take(elems * 2); q.lo = take(elems * 2); } return q; };
    w.xin = take(Bpad * h->Dpad * esz);
    w.p_xin = planes(Bpad * h->Dpad);
    w.res = (float*)take(Bpad * h->Cp * 4);
    w.loss_part = (float*)take(8192 * 4);
    w.scalar = (float*)take(256);
    if (mode == DPOSER_WS_INFER) {
        w.emb = take(Bpad * E * esz);
        w.temb = take(Bpad * E * esz);
        w.p_emb = planes(Bpad * E);
        w.p_temb = planes(Bpad * E);
        for (int i = 0; i < 3; ++i) { w.hbuf[i] = take(Bpad * H * esz); w.p_h[i] = planes(Bpad * H); }
    } else if (mode == DPOSER_WS_SHARED_T) {
        for (int i = 0; i < 3; ++i) { w.hbuf[i] = take(Bpad * H * esz); w.p_h[i] = planes(Bpad * H); }
        w.xt = (float*)take(Bpad * h->Dpad * 4);
        w.xft = (float*)take(Bpad * h->Dpad * 4);
        w.xmft = (float*)take(Bpad * h->Dpad * 4);
        w.npad = round_up(n_steps < 1 ? 1 : n_steps, 32);
        w.tt_labels = (float*)take(w.npad * 4);
        w.tt_emb = (float*)take(w.npad * E * 4);
        w.tt_temb = (float*)take(w.npad * E * 4);
        w.table = (float*)take(w.npad * (int64_t)L * H * 4);
        w.tt_t = (float*)take(w.npad * 4);
        w.smp_layers = (SamplerLayer*)take(MAX_L * sizeof(SamplerLayer));
        w.smp_sync = (uint32_t*)take((SAMPLER_CTRL_WORDS + Bpad / 256 + 1) * 4);
    } else {
        w.emb = take(Bpad * E * esz);
        w.temb = take(Bpad * E * esz);
        w.upre = take(Bpad * E * esz);
        w.p_emb = planes(Bpad * E);
        w.p_temb = planes(Bpad * E);
        w.p_dU = planes(Bpad * E);
        w.p_dres = planes(Bpad * h->Cp);
        for (int l = 0; l < L; ++l) { w.p_h[l] = planes(Bpad * H); w.p_dy[l] = planes(Bpad * H); }
        for (int l = 0; l < L; ++l) {
            w.hbuf[l] = take(Bpad * H * esz);
            w.xhat[l] = take(Bpad * H * esz);
            w.dy[l] = take(Bpad * H * esz);
            w.aux[l] = (GnAux*)take((Bpad / 32) * (H / 32) * 64 * (int64_t)(h->gs == 32 ? sizeof(GnAux) : sizeof(GnAuxG)));
            w.gn_part[l] = (float*)take((Bpad / 32) * 3 * (int64_t)H * 4);
        }
        w.carry[0] = take(Bpad * H * esz);
        w.carry[1] = take(Bpad * H * esz);
        w.dU = take(Bpad * E * esz);
        w.dU_part = (Bpad <= SILU_SPLIT_CAP || h->x3) ? (float*)take((int64_t)L * Bpad * E * 4) : nullptr;      // k-split partials of the time-branch dgrad (small batches; bf16x3: one GEMM per layer)
        w.dres = take(Bpad * h->Cp * esz);
        w.tbuf = (float*)take(Bpad * 4);
        w.zbuf = (float*)take(Bpad * h->Dpad * 4);
        if (!wgrad_tr_mode(h, Bpad)) {   // transposed operand copies of the wgrad GEMMs (fp32 mode only: bf16 reads sample-major)
            for (int l = 0; l < L; ++l) w.dyT[l] = take(Bpad * H * esz);
            for (int l = 0; l < L; ++l) w.hT[l] = take(Bpad * H * esz);
            w.tembT = take(Bpad * E * esz);
            w.embT = take(Bpad * E * esz);
            w.xinT = take(Bpad * h->Dpad * esz);
            w.dresT = take(Bpad * h->Cp * esz);
            w.dUT = take(Bpad * E * esz);
        }
        const int64_t nchunks = ceil_div(Bpad, 2048);
        w.cs_part_post = (float*)take(cs_post_rows(Bpad) * h->Cp * 4);      // rows: k_colsum chunks, k_dsm's blocks (<= 1024) or EpiDsm's wave rows (Bpad / 32)
        w.cs_part_se = (float*)take(nchunks * E * 4);
        w.silu_part = (float*)take(silu_part_rows(Bpad, h->x3) * (int64_t)E * 4);       // per-wave column sums of dU (time-branch dgrad epilogue), or k_silu_bwd_reduce's per-block sums
        // slabs: worst case ksplit 32 is never reached for the big tensors; size exactly below
        const int64_t stages = Bpad / (h->KBS * 4);
        int64_t slab_elems = 0;
        auto acc = [&](int n_rows_pad, int k_rows_pad, int64_t numel) {
            const int shape = wgrad_shape(n_rows_pad, k_rows_pad, Bpad);
            const int64_t tiles = (int64_t)(n_rows_pad / (shape_ct(shape) * 32)) * (k_rows_pad / (shape_st(shape) * 32));
            slab_elems += (int64_t)pick_ksplit(tiles, stages, shape == SHAPE_BIG ? 256 : 512) * numel * h->wk;      // (bf16x3: one slab set per product term)
        };
        for (int l = 0; l < L; ++l) {
            acc(H, h->layer[l].kin_pad, (int64_t)H * h->layer[l].kin);
            acc(H, E, (int64_t)H * E);
        }
        acc(h->Cp, H, (int64_t)h->D * H);
        acc(E, E, (int64_t)E * E);
        {   // room for the partial tiles of the one-launch weight gradients (wgrad_batch.h; plan_batched_wgrad has the conditions)
            const int64_t need = (int64_t)WGB_BLOCKS * WGB_MAX_SEG * 65536 * h->wk + ((int64_t)14 << 20) * h->wk;      // (bf16x3: one partial set per product term)
            if (score_tuning().wgrad_batched != 0 && (!h->f32 || h->x3) && H == 1024 && E == 512 && slab_elems < need) slab_elems = need;
        }
        w.slabs = (float*)take(slab_elems * 4);
        w.slab_elems = slab_elems;
    }
    w.total = p;
}

extern "C" int64_t dposer_scorefc_workspace_bytes(dposer_scorefc_t h, int64_t batch, int32_t mode, int32_t n_steps) {
    if (!h || batch <= 0) return -1;
    Ws w;
    layout_ws(h, batch, mode, n_steps, nullptr, w);
    return w.total;
}

// ------------------------------------------------------------------------------------------------
// GEMM launch helpers
// ------------------------------------------------------------------------------------------------
static thread_local double g_next_flops = 0.0;   // algorithmic FLOPs of the next GEMM launch (profiling only)
static GemmArgs gemm_args(const void* W, int w_stride_blocks, int n_cblk, int n_sblk) {
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.alg_flops = g_next_flops;
    g_next_flops = 0.0;
    g.W = W;
    g.w_stride_blocks = w_stride_blocks;
    g.n_cblk = n_cblk;
    g.n_sblk = n_sblk;
    g.ksplit = 1;
    return g;
}
static void add_seg(GemmArgs& g, const void* src, int kblocks) {
    g.src[g.nseg] = src;
    g.seg_kblocks[g.nseg] = kblocks;
    g.nseg++;
    g.ktot_blocks += kblocks;
}

// ---- bf16x3 mode: GEMM operands are the bf16 planes of the stored fp32 activations --------------------------------------------------
static inline int gemm_prec(const dposer_scorefc_s* h) { return h->x3 ? PREC_BF16X3 : (h->w32 ? PREC_FP32 : PREC_BF16); }
// the activation operand `stored` ([Bpad][kblocks * KBS]) of a GEMM: one segment, or its planes as the three segments (hi, hi, lo) that
// meet the packed weight columns [hi | lo | hi]
static void add_act(const dposer_scorefc_s* h, GemmArgs& g, const void* stored, const Planes& pl, int kblocks) {
    if (!h->x3) { add_seg(g, stored, kblocks); return; }
    add_seg(g, pl.hi, kblocks);
    add_seg(g, pl.hi, kblocks);
    add_seg(g, pl.lo, kblocks);
}
// (re)build the planes of a stored activation [Bpad][K] (no-op outside bf16x3 mode)
static int split_act(const dposer_scorefc_s* h, const void* stored, const Planes& pl, int64_t Bpad, int K, hipStream_t st) {
    if (!h->x3) return DPOSER_OK;
    DP_HIP_LAUNCH(launch_split_ft32(stored, pl.hi, pl.lo, Bpad, K, st));
    return DPOSER_OK;
}

static thread_local int64_t g_alg_batch = 0;   // un-padded batch of the running call (profiling only)
static DropoutCfg drop_cfg(const dposer_scorefc_s* h, bool train, int site, uint64_t seed, uint32_t step) {
    DropoutCfg d;
    std::memset(&d, 0, sizeof(d));
    if (train && h->d.dropout_p > 0.f) {
        d.p = h->d.dropout_p;
        d.scale = 1.0f / (1.0f - d.p);
        d.thr = (uint32_t)((1.0 - (double)d.p) * 65536.0);
        d.site = site;
        d.offset = step;
        d.seed = seed;
        d.groups_x4 = h->H / 8;
        if (h->dbg_keep) { d.ext_keep = h->dbg_keep + (int64_t)site * h->dbg_keep_batch * h->H; d.ext_rows = h->dbg_keep_batch; }
    }
    return d;
}
// TEST HOOK: keep decisions of every dropout site as bytes [n_layers][batch][H] on the device (NULL: back to the Philox streams).  Lets a
// test feed the masks the REFERENCE drew with torch's generator (golden g4) through the fused training step; honoured by the
// tile-per-group training epilogue (hidden_dim 1024) for calls with exactly this batch size.
extern "C" int dposer_scorefc_debug_set_dropout_masks(dposer_scorefc_t h, const unsigned char* keep, int64_t batch) {
#ifndef DPOSER_TEST_HOOKS
    (void)h; (void)batch;
    DP_CHECK_ARG(!keep, "this library was built without DPOSER_TEST_HOOKS: the training epilogues have no injected-mask branch "
                        "(load libdposer_hip_testhooks.so, DPOSER_LIB_PATH, for tests that need it)");
    return DPOSER_OK;
#endif
    DP_CHECK_ARG(h, "null handle");
    DP_CHECK_ARG(!keep || (batch > 0 && h->gs == 32), "injected dropout masks: hidden_dim 1024 only, batch > 0");
    h->dbg_keep = keep;
    h->dbg_keep_batch = keep ? batch : 0;
    return DPOSER_OK;
}

// One GroupNorm layer.  per_sample_t: K-concat(h, temb) with packed bias_cat; else x-path only with
// `bias_row` (time-table row of this step and layer).
static int run_gn_layer(dposer_scorefc_s* h, const float* flat, const char* packed, int l, const void* in, const Planes& in_pl, const void* temb,
                        const Planes& temb_pl, const float* bias_row, void* out, const Planes& out_pl, const void* resid, void* xhat, GnAux* aux,
                        bool train, int64_t Bpad, uint64_t seed, uint32_t step, hipStream_t st) {
    const int shape = gn_shape(Bpad, h->H, h->gs, !h->f32);
    const LayerOff& lo = h->layer[l];
    const int kx = lo.kin_pad / h->KBS, ke = h->E / h->KBS;
    g_next_flops = 2.0 * (double)g_alg_batch * h->H * (lo.kin + (temb ? h->E : 0));
    GemmArgs g = gemm_args(packed + h->pk_wl[l], (kx + ke) * h->wk, h->H / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
    add_act(h, g, in, in_pl, kx);
    if (temb) add_act(h, g, temb, temb_pl, ke);
    GNParams p;
    p.bias = temb ? reinterpret_cast<const float*>(packed + h->pk_bias_cat) + (int64_t)l * h->H : bias_row;
    p.gamma = flat + lo.gamma;
    p.beta = flat + lo.beta;
    p.out = out;
    p.resid = resid;
    p.xhat = xhat;
    p.aux = aux;
    p.H = h->H;
    p.outT = nullptr;
    p.Spad = Bpad;
    p.act = h->d.activation;
    p.drop = drop_cfg(h, train, l, seed, step);
    // bf16x3: the epilogue writes the operand planes of the consuming GEMMs itself; the fp32 tiles are kept only where a later layer adds
    // them as its residual input (the outputs of layers 0, 2, ... below the last block)
    p.out_hi = h->x3 ? out_pl.hi : nullptr;
    p.out_lo = h->x3 ? out_pl.lo : nullptr;
    if (h->x3 && !((l % 2) == 0 && l + 2 < h->L)) p.out = nullptr;
    DP_HIP_LAUNCH(gemm_gn(gemm_prec(h), train, shape, g, p, st, h->gs));
    return DPOSER_OK;
}

static int run_post(dposer_scorefc_s* h, const float* flat, const char* packed, const void* in, const Planes& in_pl, float* res, int64_t B, int64_t Bpad, hipStream_t st) {
    const int shape = final_shape(Bpad);
    g_next_flops = 2.0 * (double)B * h->D * h->H;
    GemmArgs g = gemm_args(packed + h->pk_wpost, h->H / h->KBS * h->wk, h->Cp / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
    add_act(h, g, in, in_pl, h->H / h->KBS);
    RowMajorParams p;
    p.bias = flat + h->off_post_b;
    p.out = res;
    p.ldc = h->Cp;
    p.C_valid = h->D;
    p.S_valid = Bpad;     // padded rows are written too (finite, never read back as samples)
    (void)B;
    DP_HIP_LAUNCH(gemm_rowmajor(gemm_prec(h), shape, g, p, st));
    return DPOSER_OK;
}

// shared_time_embed: temb = SiLU(emb W^T + b); in bf16x3 mode the planes of `emb` are built here (emb comes from an elementwise kernel) and
// those of temb behind the GEMM
static int run_temb(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, void* upre, void* tembT, bool train, hipStream_t st) {
    const int64_t Bpad = w.Bpad;
    const int shape = main_shape(Bpad, h->E);
    g_next_flops = 2.0 * (double)g_alg_batch * h->E * h->E;
    DP_TRY(split_act(h, w.emb, w.p_emb, Bpad, h->E, st));
    GemmArgs g = gemm_args(packed + h->pk_wse, h->E / h->KBS * h->wk, h->E / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
    add_act(h, g, w.emb, w.p_emb, h->E / h->KBS);
    BiasSiLUParams p;
    p.bias = flat + h->off_se_b;
    p.out = w.temb;
    p.pre = upre;
    p.N = h->E;
    p.outT = tembT;
    p.Spad = Bpad;
    p.act = h->d.activation;
    std::memset(&p.drop, 0, sizeof(p.drop));
    p.out_hi = h->x3 ? w.p_temb.hi : nullptr;      // (bf16x3: temb's operand planes straight from the epilogue)
    p.out_lo = h->x3 ? w.p_temb.lo : nullptr;
    DP_HIP_LAUNCH(gemm_bias_silu(gemm_prec(h), train, shape, g, p, st));
    return DPOSER_OK;
}

// scale_by_sigma as the shared-t kernels take it: 0 off, 1 divide by sigmas[(int)label] (positional embedding, model.py:159), 2 divide by the
// label itself (Fourier embedding: used_sigmas = t, model.py:152)
static inline int sbs_mode(const dposer_scorefc_s* h) { return h->d.scale_by_sigma ? (h->d.embedding == DPOSER_EMB_FOURIER ? 2 : 1) : 0; }
static int check_common(dposer_scorefc_t h, const float* flat, const void* packed, void* ws, int64_t batch) {
    DP_CHECK_ARG(h && flat && packed && ws, "null argument");
    g_act = h->d.activation;
    DP_CHECK_ARG(batch > 0, "batch must be positive");
    DP_CHECK_ARG(((uintptr_t)flat & 15) == 0, "flat_params must be 16-byte aligned");
    DP_CHECK_ARG(((uintptr_t)packed & 255) == 0 && ((uintptr_t)ws & 255) == 0, "packed / workspace must be 256-byte aligned");
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// ScoreModelFC.forward
// ------------------------------------------------------------------------------------------------
extern "C" int dposer_scorefc_forward(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const float* x,
                                      const float* labels, const float* freq, const float* sigmas, float* out, int64_t B,
                                      void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(x && labels && freq && sigmas && out, "null tensor argument");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_INFER, 0, (char*)ws_, w);
    PrepArgs pa;
    pa.x = x; pa.labels = labels; pa.freq = freq; pa.xin = w.xin; pa.emb = w.emb;
    pa.B = B; pa.Bpad = w.Bpad; pa.D = h->D; pa.Dpad = h->Dpad; pa.E = h->E;
    pa.fourier = h->d.embedding == DPOSER_EMB_FOURIER; pa.f32 = h->f32;
    DP_HIP_LAUNCH(launch_prep_infer(pa, st));
    DP_TRY(run_temb(h, flat, packed, w, nullptr, nullptr, false, st));
    DP_TRY(split_act(h, w.xin, w.p_xin, w.Bpad, h->Dpad, st));
    const void* in = w.xin;
    const Planes* in_pl = &w.p_xin;
    for (int l = 0; l < h->L; ++l) {
        void* o = w.hbuf[l % 3];
        const void* resid = (l >= 2 && (l % 2) == 0) ? w.hbuf[(l - 2) % 3] : nullptr;
        DP_TRY(run_gn_layer(h, flat, packed, l, in, *in_pl, w.temb, w.p_temb, nullptr, o, w.p_h[l % 3], resid, nullptr, nullptr, false, w.Bpad, 0, 0, st));
        in = o;
        in_pl = &w.p_h[l % 3];
    }
    DP_TRY(run_post(h, flat, packed, in, *in_pl, w.res, B, w.Bpad, st));
    OutModelArgs oa;
    oa.res = w.res; oa.labels = labels; oa.sigmas = sigmas; oa.out = out; oa.B = B; oa.D = h->D; oa.Cp = h->Cp;
    oa.num_scales = h->d.num_scales; oa.scale_by_sigma = h->d.scale_by_sigma; oa.fourier = pa.fourier;
    DP_HIP_LAUNCH(launch_out_model(oa, st));
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// shared-t machinery: time-bias table for a list of labels
// ------------------------------------------------------------------------------------------------
static int build_time_table(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, const float* labels_dev, float label0,
                            int64_t n, const float* freq, hipStream_t st) {
    const int E = h->E, H = h->H, L = h->L;
    const int64_t npad = round_up(n, 32);
    DP_HIP_LAUNCH(launch_time_embed(labels_dev, label0, n, npad, freq, E, h->d.embedding == DPOSER_EMB_FOURIER, w.tt_emb, st));
    const int shape = npad % 128 == 0 ? SHAPE_MID : SHAPE_SMALL;
    {
        GemmArgs g = gemm_args(packed + h->pk_wse32, E / 8, E / (shape_ct(shape) * 32), (int)(npad / (shape_st(shape) * 32)));
        add_seg(g, w.tt_emb, E / 8);
        BiasSiLUParams p;
        p.bias = flat + h->off_se_b; p.out = w.tt_temb; p.pre = nullptr; p.N = E; p.outT = nullptr; p.Spad = npad; p.act = h->d.activation;
        p.out_hi = nullptr; p.out_lo = nullptr;
        std::memset(&p.drop, 0, sizeof(p.drop));
        DP_HIP_LAUNCH(gemm_bias_silu(PREC_FP32, false, shape, g, p, st));
    }
    {
        GemmArgs g = gemm_args(packed + h->pk_wt_all32, E / 8, L * H / (shape_ct(shape) * 32), (int)(npad / (shape_st(shape) * 32)));
        add_seg(g, w.tt_temb, E / 8);
        RowMajorParams p;
        p.bias = reinterpret_cast<const float*>(packed + h->pk_bias_cat);
        p.out = w.table; p.ldc = (int64_t)L * H; p.C_valid = L * H; p.S_valid = npad;
        DP_HIP_LAUNCH(gemm_rowmajor(PREC_FP32, shape, g, p, st));
    }
    return DPOSER_OK;
}

// the GroupNorm layers of one shared-t network evaluation (bias rows from table row `row`); returns the last activation (and its planes).
// bf16x3: the planes of the input `xin` are built here (every producer of xin is an elementwise kernel or an fp32-storing epilogue).
static int run_shared_t_layers(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, int64_t row, hipStream_t st, const void** last,
                               const Planes** last_pl) {
    DP_TRY(split_act(h, w.xin, w.p_xin, w.Bpad, h->Dpad, st));
    const void* in = w.xin;
    const Planes* in_pl = &w.p_xin;
    const float* trow = w.table + row * (int64_t)h->L * h->H;
    const Planes none = {nullptr, nullptr};
    for (int l = 0; l < h->L; ++l) {
        void* o = w.hbuf[l % 3];
        const void* resid = (l >= 2 && (l % 2) == 0) ? w.hbuf[(l - 2) % 3] : nullptr;
        DP_TRY(run_gn_layer(h, flat, packed, l, in, *in_pl, nullptr, none, trow + (int64_t)l * h->H, o, w.p_h[l % 3], resid, nullptr, nullptr, false, w.Bpad, 0, 0, st));
        in = o;
        in_pl = &w.p_h[l % 3];
    }
    *last = in;
    *last_pl = in_pl;
    return DPOSER_OK;
}

// one shared-t network evaluation: xin -> res, bias rows from table row `row`
static int run_shared_t(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, int64_t row, int64_t B, hipStream_t st) {
    const void* last = nullptr;
    const Planes* last_pl = nullptr;
    DP_TRY(run_shared_t_layers(h, flat, packed, w, row, st, &last, &last_pl));
    return run_post(h, flat, packed, last, *last_pl, w.res, B, w.Bpad, st);
}

static SdeCfg to_sde(const dposer_sde_desc* s) {
    SdeCfg c;
    c.kind = (s->kind == DPOSER_SDE_VP || s->kind == DPOSER_SDE_VP_DISCRETE) ? SDE_VP : ((s->kind == DPOSER_SDE_VE || s->kind == DPOSER_SDE_VE_DISCRETE) ? SDE_VE : SDE_SUBVP);
    c.discrete = s->kind == DPOSER_SDE_VE_DISCRETE || s->kind == DPOSER_SDE_VP_DISCRETE;
    c.beta_0 = s->beta_min;
    c.beta_1 = s->beta_max;
    c.N = s->N;
    c.T = (float)s->T;
    return c;
}

// (shared-t entry points; the training step takes the continuous kinds only: dsm_loss_fwd_bwd_impl)
static bool sde_kind_ok(const dposer_sde_desc* s) {
    return s->kind == DPOSER_SDE_SUBVP || s->kind == DPOSER_SDE_VP || s->kind == DPOSER_SDE_VE || s->kind == DPOSER_SDE_VE_DISCRETE || s->kind == DPOSER_SDE_VP_DISCRETE;
}
// What the network is conditioned on at the n step times `t_host`, into w.tt_labels: t * 999 (utils.py:152: one IEEE fp32 product, the same
// bits on host and device) or, for VE, sigma(t) (utils.py:173) -- formed ON THE DEVICE from the staged times (k_ve_labels), because every
// kernel that perturbs with sigma(t) / divides the output by it calls the device's powf, and the host's libm may round the last bit the other way.
static int stage_step_labels(dposer_scorefc_s* h, Ws& w, const dposer_sde_desc* s, const float* t_host, int n, hipStream_t st) {
    const bool ve = s && s->kind == DPOSER_SDE_VE;
    const bool ve_disc = s && s->kind == DPOSER_SDE_VE_DISCRETE;      // round((T - t)(N - 1)): exact fp32 operations, formed here
    const bool vp_disc = s && s->kind == DPOSER_SDE_VP_DISCRETE;      // t (N - 1), utils.py:158
    h->host_stage.resize(n);
    for (int i = 0; i < n; ++i)
        h->host_stage[i] = ve ? t_host[i] : (ve_disc ? sde_ve_discrete_label((float)s->T, (float)(s->N - 1), t_host[i])
                                                     : (vp_disc ? t_host[i] * (float)(s->N - 1) : t_host[i] * 999.0f));
    DP_CHECK_HIP(hipMemcpyAsync(w.tt_labels, h->host_stage.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
    if (ve) {
        const SdeDev d = make_sde_dev(to_sde(s));
        DP_HIP_LAUNCH(launch_ve_labels(w.tt_labels, n, d.smin, d.ratio, st));
    }
    return DPOSER_OK;
}
// the one-row time table of a call at a single time t
static int build_time_table_at(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, const dposer_sde_desc* s, float t, const float* freq,
                               hipStream_t st) {
    if (s && s->kind == DPOSER_SDE_VE_DISCRETE) return build_time_table(h, flat, packed, w, nullptr, sde_ve_discrete_label((float)s->T, (float)(s->N - 1), t), 1, freq, st);
    if (s && s->kind == DPOSER_SDE_VP_DISCRETE) return build_time_table(h, flat, packed, w, nullptr, t * (float)(s->N - 1), 1, freq, st);
    if (!s || s->kind != DPOSER_SDE_VE) return build_time_table(h, flat, packed, w, nullptr, t * 999.0f, 1, freq, st);      // the label travels by value
    DP_TRY(stage_step_labels(h, w, s, &t, 1, st));
    return build_time_table(h, flat, packed, w, w.tt_labels, 0.f, 1, freq, st);
}
static int em_sampler_impl(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde, float* x,
                           float* x_mean, const float* timesteps_host, int32_t start_step, int32_t n_steps, const float* observation,
                           const float* mask, const float* noise, uint64_t seed, float* traj, int32_t traj_stride, const float* freq,
                           const float* sigmas, int64_t B, void* stream);
extern "C" int dposer_em_sampler(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                 float* x, float* x_mean, const float* timesteps_host, int32_t start_step, const float* observation,
                                 const float* mask, const float* noise, uint64_t seed, float* traj, int32_t traj_stride,
                                 const float* freq, const float* sigmas, int64_t B, void* stream) {
    return em_sampler_impl(h, flat, packed_, ws_, sde, x, x_mean, timesteps_host, start_step, -1, observation, mask, noise, seed, traj,
                           traj_stride, freq, sigmas, B, stream);
}
extern "C" int dposer_em_sampler_steps(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                       float* x, float* x_mean, const float* timesteps_host, int32_t start_step, int32_t n_steps,
                                       const float* observation, const float* mask, const float* noise, uint64_t seed, float* traj,
                                       int32_t traj_stride, const float* freq, const float* sigmas, int64_t B, void* stream) {
    DP_CHECK_ARG(n_steps >= 0, "n_steps must be >= 0");
    return em_sampler_impl(h, flat, packed_, ws_, sde, x, x_mean, timesteps_host, start_step, n_steps, observation, mask, noise, seed, traj,
                           traj_stride, freq, sigmas, B, stream);
}
static int em_sampler_impl(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde, float* x,
                           float* x_mean, const float* timesteps_host, int32_t start_step, int32_t n_steps, const float* observation,
                           const float* mask, const float* noise, uint64_t seed, float* traj, int32_t traj_stride, const float* freq,
                           const float* sigmas, int64_t B, void* stream) {
    DpRange _dp_range("dposer_em_sampler");
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(sde && x && x_mean && timesteps_host && freq && sigmas, "null argument");
    DP_CHECK_ARG(sde_kind_ok(sde), "unknown SDE kind");
    DP_CHECK_ARG(sde->N >= 1 && start_step >= 0 && start_step <= sde->N, "bad step range");
    DP_CHECK_ARG((observation == nullptr) == (mask == nullptr), "observation and mask go together");
    DP_CHECK_ARG(traj_stride >= 1, "traj_stride must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    const int N = sde->N;
    const int n_run = (n_steps < 0 || n_steps > N - start_step) ? N - start_step : n_steps;   // (a range ends without look-ahead imputation)
    if (n_run == 0) return DPOSER_OK;
    Ws w;
    layout_ws(h, B, DPOSER_WS_SHARED_T, n_run, (char*)ws_, w);
    // labels = t * 999 (utils.py:152; VE: sigma(t), on the device), one H2D copy
    DP_TRY(stage_step_labels(h, w, sde, timesteps_host + start_step, n_run, st));
    DP_TRY(build_time_table(h, flat, packed, w, w.tt_labels, 0.f, n_run, freq, st));

    const SdeCfg sc = to_sde(sde);
    const int k_noise = observation ? 3 : 1;
    const int64_t BD = B * h->D;
    EmUpdateArgs ea;
    std::memset(&ea, 0, sizeof(ea));
    ea.x = x; ea.x_mean = x_mean; ea.xin = w.xin; ea.sigmas = sigmas; ea.obs = observation; ea.mask = mask;
    ea.B = B; ea.Bpad = w.Bpad; ea.D = h->D; ea.Dpad = h->Dpad; ea.Cp = h->Cp; ea.num_scales = h->d.num_scales;
    ea.f32 = h->f32; ea.scale_by_sigma = sbs_mode(h); ea.sde = sc; ea.seed = seed;
    // step "-1": imputation ahead of the first predictor call (sampling.py:459) + pack x
    ea.res = nullptr; ea.t = timesteps_host[start_step]; ea.t_next = timesteps_host[start_step];
    ea.step = (uint32_t)(start_step - 1);
    ea.z_impA = (noise && observation) ? noise : nullptr;
    // Fast path (plain generation: no observation, in-kernel noise, no trajectory): the state stays in HBM as fp32 FT and
    // post_dense + the Euler-Maruyama update are ONE GEMM launch per step (EpiEmStep) -- no `res` round trip, no update kernel.
    const bool fused = !observation && !noise && !traj && h->Cp == h->Dpad;
    if (fused) ea.x_ft = w.xft;
    DP_HIP_LAUNCH(launch_em_update(ea, st));
    // Persistent form of the fast path (gemm_sampler.hip): one workgroup per 256 samples walks every layer of every step -- no
    // grid-wide join per layer, no launch per layer.  Same kernels' code per tile: bit-identical samples (tested).  Needs the
    // 256 x 256 tiling (hidden_dim 1024 groups, batch padded to 256) and the 64-channel post_dense tile.
    // OFF by default (DPOSER_SAMPLER_PERSISTENT=1 enables it): measured (tools/sampler_ab.py, profiles/r03_sampler_ab.txt)
    //   65536 samples: 715 us per step against 680 us for the six launches;  32768 / 16384 samples: 453 us against 341 / 182 us.
    // A workgroup alone on its CU needs 453 us per step (26 us per 256 x 256 x 1024 tile) however few of them run -- so below 256
    // blocks the launches win by using every CU -- and at 256 blocks the same work takes 715 us: a workgroup re-reads its 512 KB
    // input panel for each of the 4 channel tiles (the launches let 4 tiles on one XCD share it through L2), 3.3 GB per step
    // through MALL / HBM, under the chip's power cap.  Removing the grid-wide joins does not pay for that.
    const int persistent_env = score_tuning().sampler_persistent;
    const int64_t persistent_min = score_tuning().sampler_persistent_min;
    if (fused && persistent_env && !(sc.kind == SDE_VP && sc.discrete) && !h->x3 && h->d.activation == DPOSER_ACT_SWISH && h->gs == 32 && h->H % 256 == 0 && h->Cp == 64 && w.Bpad % 256 == 0 && (persistent_env >= 2 || w.Bpad >= persistent_min)) {
        SamplerLayer tab[MAX_L];
        std::memset(tab, 0, sizeof(tab));
        for (int l = 0; l < h->L; ++l) {
            const LayerOff& lo = h->layer[l];
            tab[l].W = packed + h->pk_wl[l];
            tab[l].in = l == 0 ? (const void*)w.xin : (const void*)w.hbuf[(l - 1) % 3];
            tab[l].resid = (l >= 2 && (l % 2) == 0) ? w.hbuf[(l - 2) % 3] : nullptr;
            tab[l].out = w.hbuf[l % 3];
            tab[l].gamma = flat + lo.gamma;
            tab[l].beta = flat + lo.beta;
            tab[l].w_stride_blocks = (lo.kin_pad + h->E) / h->KBS;
            tab[l].kblocks = lo.kin_pad / h->KBS;
        }
        DP_CHECK_HIP(hipMemcpyAsync(w.smp_layers, tab, sizeof(SamplerLayer) * h->L, hipMemcpyHostToDevice, st));
        DP_CHECK_HIP(hipMemcpyAsync(w.tt_t, timesteps_host + start_step, n_run * sizeof(float), hipMemcpyHostToDevice, st));
        // (both copies come from host memory that must stay valid until they have executed: pageable -> the runtime stages them
        //  before returning; `tab` lives on this stack frame)
        SamplerArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.layers = w.smp_layers; sa.L = h->L; sa.H = h->H; sa.Spad = w.Bpad; sa.table = w.table; sa.tsteps = w.tt_t; sa.n_steps = n_run;
        sa.step0 = (uint32_t)start_step; sa.Wpost = packed + h->pk_wpost; sa.post_kblocks = h->H / h->KBS; sa.last = w.hbuf[(h->L - 1) % 3];
        sa.x_mean_ft = w.xmft;
        EmStepParams& p = sa.em;
        p.bias = flat + h->off_post_b; p.x_ft = w.xft; p.x_mean_ft = nullptr; p.xin = w.xin;
        p.sigmas = sigmas; p.sde = make_sde_dev(sc); p.t = 0.f; p.num_scales = h->d.num_scales;
        p.scale_by_sigma = sbs_mode(h); p.D = h->D; p.Cp = h->Cp; p.QD = (h->D + 3) >> 2; p.S_valid = B;
        p.seed = seed; p.step = 0;
        if (persistent_env >= 2) {
            // cluster form (gemm_sampler.hip): 2 = joined by the per-block counters, 3 = the same walk without waits (timing probe, garbage samples)
            sa.n_sblk = (int)(w.Bpad / 256);
            sa.ctrl = w.smp_sync; sa.progress = w.smp_sync + SAMPLER_CTRL_WORDS;
            DP_CHECK_HIP(hipMemsetAsync(w.smp_sync, 0, (SAMPLER_CTRL_WORDS + sa.n_sblk) * 4, st));
            DP_HIP_LAUNCH(launch_sampler_cluster(h->f32 ? PREC_FP32 : PREC_BF16, sa, persistent_env == 2, st));
            DP_HIP_LAUNCH(launch_ft_to_rows(w.xft, x, w.xmft, x_mean, B, w.Bpad, h->D, h->Dpad, st));
            uint32_t ctrl[SAMPLER_CTRL_WORDS];
            DP_CHECK_HIP(hipMemcpyAsync(ctrl, w.smp_sync, sizeof(ctrl), hipMemcpyDeviceToHost, st));
            DP_CHECK_HIP(hipStreamSynchronize(st));
            if (getenv("DPOSER_SAMPLER_CLUSTER_REPORT"))
                fprintf(stderr, "[dposer] cluster sampler: workgroups per XCD %u %u %u %u %u %u %u %u, error %u, longest wait %u polls\n",
                        ctrl[0], ctrl[1], ctrl[2], ctrl[3], ctrl[4], ctrl[5], ctrl[6], ctrl[7], ctrl[8], ctrl[9]);
            if (ctrl[8]) return dposer_set_error(DPOSER_ERR_HIP, std::string("cluster sampler: ") + (ctrl[8] == 2 ? "workgroups not dealt evenly to the XCDs" : "a workgroup waited past its poll budget (not all resident?)"));
            return DPOSER_OK;
        }
        DP_HIP_LAUNCH(launch_sampler_persistent(h->f32 ? PREC_FP32 : PREC_BF16, sa, w.Bpad / 256, st));
        DP_HIP_LAUNCH(launch_ft_to_rows(w.xft, x, w.xmft, x_mean, B, w.Bpad, h->D, h->Dpad, st));
        return DPOSER_OK;
    }
    if (fused) {
        const int shape = final_shape(w.Bpad);
        for (int i = 0; i < n_run; ++i) {
            const int gi = start_step + i;
            const void* last = nullptr;
            const Planes* last_pl = nullptr;
            DP_TRY(run_shared_t_layers(h, flat, packed, w, i, st, &last, &last_pl));
            g_next_flops = 2.0 * (double)B * h->D * h->H;
            GemmArgs g = gemm_args(packed + h->pk_wpost, h->H / h->KBS * h->wk, h->Cp / (shape_ct(shape) * 32), (int)(w.Bpad / (shape_st(shape) * 32)));
            add_act(h, g, last, *last_pl, h->H / h->KBS);
            EmStepParams p;
            std::memset(&p, 0, sizeof(p));
            p.bias = flat + h->off_post_b; p.x_ft = w.xft; p.x_mean_ft = (i + 1 == n_run) ? w.xmft : nullptr; p.xin = w.xin;
            p.sigmas = sigmas; p.sde = make_sde_dev_at(sc, timesteps_host[gi]); p.t = timesteps_host[gi]; p.num_scales = h->d.num_scales;
            p.scale_by_sigma = sbs_mode(h); p.D = h->D; p.Cp = h->Cp; p.QD = (h->D + 3) >> 2; p.S_valid = B;
            p.seed = seed; p.step = (uint32_t)gi;
            DP_HIP_LAUNCH(gemm_em_step(gemm_prec(h), shape, g, p, st));
        }
        DP_HIP_LAUNCH(launch_ft_to_rows(w.xft, x, w.xmft, x_mean, B, w.Bpad, h->D, h->Dpad, st));
        return DPOSER_OK;
    }
    for (int i = 0; i < n_run; ++i) {
        const int gi = start_step + i;
        DP_TRY(run_shared_t(h, flat, packed, w, i, B, st));
        ea.res = w.res;
        ea.t = timesteps_host[gi];
        ea.t_next = (i + 1 < n_run) ? timesteps_host[gi + 1] : -1.0f;
        ea.step = (uint32_t)gi;
        const float* nz = noise ? noise + (int64_t)i * k_noise * BD : nullptr;
        ea.z_pred = nz ? nz + (observation ? BD : 0) : nullptr;
        ea.z_impB = (nz && observation) ? nz + 2 * BD : nullptr;
        ea.z_impA = (nz && observation && i + 1 < n_run) ? nz + (int64_t)k_noise * BD : nullptr;
        ea.traj = (traj && ((i + 1) % traj_stride == 0)) ? traj + (int64_t)((i + 1) / traj_stride - 1) * BD : nullptr;
        DP_HIP_LAUNCH(launch_em_update(ea, st));
    }
    return DPOSER_OK;
}

// LangevinCorrector.update_fn (sampling.py:282-302), one corrector step at a shared t, in two phases around the batch means:
//   phase 0: pack x, evaluate the network, write sum_b ||grad_b|| and sum_b ||noise_b|| of THIS rank's samples to norm_sums[0..1]
//   (the caller all-reduces the two floats over the data-parallel ranks -- or not, on one GPU)
//   phase 1: x_mean = x + step * grad, x = x_mean + sqrt(2 step) * noise with step from norm_sums * inv_global_batch.
// The workspace keeps the network output between the two phases: nothing else may run on it in between.
extern "C" int dposer_langevin_step(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                    float* x, float* x_mean, float t, float alpha, float snr, const float* noise, uint64_t seed,
                                    uint32_t step, float* norm_sums, int32_t phase, double inv_global_batch, const float* freq,
                                    const float* sigmas, int64_t B, void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(sde && x && norm_sums && freq && sigmas, "null argument");
    DP_CHECK_ARG(sde_kind_ok(sde), "unknown SDE kind");
    DP_CHECK_ARG(phase == 0 || phase == 1, "phase must be 0 (norms) or 1 (update)");
    DP_CHECK_ARG(phase == 0 || x_mean, "x_mean is required in the update phase");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_SHARED_T, 1, (char*)ws_, w);
    LangevinArgs a;
    std::memset(&a, 0, sizeof(a));
    a.res = w.res; a.noise = noise; a.sigmas = sigmas; a.x = x; a.x_mean = x_mean; a.part = w.loss_part; a.norm_sums = norm_sums;
    a.t = t; a.alpha = alpha; a.snr = snr; a.inv_global_batch = (float)inv_global_batch; a.B = B; a.Bpad = w.Bpad; a.D = h->D; a.Dpad = h->Dpad;
    a.Cp = h->Cp; a.num_scales = h->d.num_scales; a.scale_by_sigma = sbs_mode(h); a.f32 = h->f32; a.sde = to_sde(sde); a.seed = seed;
    a.step = step;
    if (phase == 0) {
        DP_TRY(build_time_table_at(h, flat, packed, w, sde, t, freq, st));
        DP_HIP_LAUNCH(launch_pack_rows(x, w.xin, B, w.Bpad, h->D, h->Dpad, h->f32, st));
        DP_TRY(run_shared_t(h, flat, packed, w, 0, B, st));
        int nb = 0;
        DP_HIP_LAUNCH(launch_langevin_norms(a, &nb, st));
        DP_HIP_LAUNCH(launch_sum_partials2(w.loss_part, nb, norm_sums, st));
    } else {
        DP_HIP_LAUNCH(launch_langevin_update(a, st));
    }
    return DPOSER_OK;
}

// table_rows <= 0: the time-bias row is built for this call (one row).  table_rows > 0: row `row` of a table that
// dposer_prior_table_build wrote into the same workspace (laid out for table_rows rows) -- the task loops build it once.
static int prior_loss_impl(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                           const float* x0, const float* z, float t, int32_t weighted, float inv_n, float* x0_hat, float* grad,
                           float* loss, uint64_t seed, uint32_t step, const float* freq, const float* sigmas, int64_t B, void* stream,
                           int32_t row, int32_t table_rows) {
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(sde && x0 && loss && sigmas && (freq || table_rows > 0), "null argument");
    DP_CHECK_ARG(sde_kind_ok(sde), "unknown SDE kind");
    DP_CHECK_ARG(table_rows <= 0 || (row >= 0 && row < table_rows), "table row out of range");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_SHARED_T, table_rows > 0 ? table_rows : 1, (char*)ws_, w);
    if (table_rows <= 0) DP_TRY(build_time_table_at(h, flat, packed, w, sde, t, freq, st));
    const SdeCfg sc = to_sde(sde);
    PerturbSharedArgs pa;
    pa.x0 = x0; pa.z_in = z; pa.xin = w.xin; pa.xt = w.xt; pa.t = t; pa.B = B; pa.Bpad = w.Bpad; pa.D = h->D; pa.Dpad = h->Dpad;
    pa.f32 = h->f32; pa.sde = sc; pa.seed = seed; pa.step = step;
    DP_HIP_LAUNCH(launch_perturb_shared(pa, st));
    DP_TRY(run_shared_t(h, flat, packed, w, table_rows > 0 ? row : 0, B, st));
    DenoiseArgs da;
    da.res = w.res; da.x0 = x0; da.xt = w.xt; da.sigmas = sigmas; da.x0_hat = x0_hat; da.grad = grad; da.loss_part = w.loss_part;
    da.t = t; da.inv_n = inv_n; da.weighted = weighted; da.B = B; da.D = h->D; da.Dpad = h->Dpad; da.Cp = h->Cp;
    da.num_scales = h->d.num_scales; da.scale_by_sigma = sbs_mode(h); da.sde = sc;
    int nb = 0;
    DP_HIP_LAUNCH(launch_denoise(da, &nb, st));
    DP_HIP_LAUNCH(launch_sum_partials(w.loss_part, nb, loss, st));
    return DPOSER_OK;
}
extern "C" int dposer_prior_loss(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                 const float* x0, const float* z, float t, int32_t weighted, float inv_n, float* x0_hat, float* grad,
                                 float* loss, uint64_t seed, uint32_t step, const float* freq, const float* sigmas, int64_t B,
                                 void* stream) {
    DP_RANGE();
    return prior_loss_impl(h, flat, packed_, ws_, sde, x0, z, t, weighted, inv_n, x0_hat, grad, loss, seed, step, freq, sigmas, B, stream, 0, 0);
}
extern "C" int dposer_prior_table_build(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const float* t_host,
                                        int32_t n_rows, const float* freq, int64_t B, void* stream) {
    return dposer_prior_table_build_sde(h, flat, packed_, ws_, nullptr, t_host, n_rows, freq, B, stream);
}
extern "C" int dposer_prior_table_build_sde(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                            const float* t_host, int32_t n_rows, const float* freq, int64_t B, void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    DP_CHECK_ARG(t_host && freq && n_rows >= 1, "bad argument");
    DP_CHECK_ARG(!sde || sde_kind_ok(sde), "unknown SDE kind");
    hipStream_t st = (hipStream_t)stream;
    Ws w;
    layout_ws(h, B, DPOSER_WS_SHARED_T, n_rows, (char*)ws_, w);
    DP_TRY(stage_step_labels(h, w, sde, t_host, n_rows, st));      // labels = t * 999 (utils.py:152); VE: sigma(t)
    return build_time_table(h, flat, (const char*)packed_, w, w.tt_labels, 0.f, n_rows, freq, st);
}
extern "C" int dposer_prior_loss_tabled(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                        const float* x0, const float* z, float t, int32_t row, int32_t table_rows, int32_t weighted,
                                        float inv_n, float* x0_hat, float* grad, float* loss, uint64_t seed, uint32_t step,
                                        const float* sigmas, int64_t B, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(table_rows >= 1, "table_rows must be >= 1");
    return prior_loss_impl(h, flat, packed_, ws_, sde, x0, z, t, weighted, inv_n, x0_hat, grad, loss, seed, step, nullptr, sigmas, B, stream, row,
                           table_rows);
}

// DPoserComp.optimize (run/completion.py:167-207): the whole optimisation loop in one call.  Per step: perturb x at the step's
// shared t, one forward-only network evaluation on the time-table row of that step (the table for ALL steps is built once),
// and one kernel for Tweedie estimate + loss gradients + per-sample Adam.  Nothing returns to the host in between.
extern "C" int dposer_completion_optimize(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                          float* x, const float* observation, const float* mask, float* adam_m, float* adam_v,
                                          const float* t_host, const int32_t* weighted_host, const float* w_prior_host,
                                          const float* w_data_host, int32_t n_steps, double lr, double beta1, double beta2, double eps,
                                          const float* noise, uint64_t seed, uint32_t step0, const float* freq, const float* sigmas,
                                          int64_t B, void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(sde && x && observation && mask && adam_m && adam_v && t_host && weighted_host && w_prior_host && w_data_host && freq && sigmas,
                 "null argument");
    DP_CHECK_ARG(sde_kind_ok(sde), "unknown SDE kind");
    DP_CHECK_ARG(n_steps >= 0, "n_steps must be >= 0");
    if (n_steps == 0) return DPOSER_OK;
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_SHARED_T, n_steps, (char*)ws_, w);
    DP_TRY(stage_step_labels(h, w, sde, t_host, n_steps, st));      // labels = t * 999 (utils.py:152) / sigma(t) (VE, :173)
    DP_TRY(build_time_table(h, flat, packed, w, w.tt_labels, 0.f, n_steps, freq, st));
    const SdeCfg sc = to_sde(sde);
    const int64_t BD = B * h->D;
    for (int i = 0; i < n_steps; ++i) {
        PerturbSharedArgs pa;
        pa.x0 = x; pa.z_in = noise ? noise + (int64_t)i * BD : nullptr; pa.xin = w.xin; pa.xt = w.xt; pa.t = t_host[i]; pa.B = B; pa.Bpad = w.Bpad;
        pa.D = h->D; pa.Dpad = h->Dpad; pa.f32 = h->f32; pa.sde = sc; pa.seed = seed; pa.step = step0 + (uint32_t)i;
        DP_HIP_LAUNCH(launch_perturb_shared(pa, st));
        DP_TRY(run_shared_t(h, flat, packed, w, i, B, st));
        CompletionUpdateArgs ua;
        ua.res = w.res; ua.xt = w.xt; ua.obs = observation; ua.mask = mask; ua.sigmas = sigmas; ua.x = x; ua.m = adam_m; ua.v = adam_v;
        ua.t = t_host[i]; ua.inv_n = (float)(1.0 / ((double)B * (double)h->D)); ua.w_prior = w_prior_host[i]; ua.w_data = w_data_host[i];
        ua.weighted = weighted_host[i];
        const double k = (double)(i + 1);
        ua.step_size = (float)(lr / (1.0 - std::pow(beta1, k)));                   // torch: step_size = lr / bias_correction1
        ua.one_minus_beta1 = (float)(1.0 - beta1); ua.beta2 = (float)beta2; ua.one_minus_beta2 = (float)(1.0 - beta2);
        ua.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(beta2, k)); ua.eps = (float)eps;
        ua.B = B; ua.D = h->D; ua.Dpad = h->Dpad; ua.Cp = h->Cp; ua.num_scales = h->d.num_scales; ua.scale_by_sigma = sbs_mode(h); ua.sde = sc;
        DP_HIP_LAUNCH(launch_completion_update(ua, st));
    }
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// training: DSM loss forward + backward
// ------------------------------------------------------------------------------------------------
// dW = dy^T @ in over the batch.  dyT / inT: the operands transposed (FT [rows][Bpad]); dy / in: the same matrices sample-major
// (FT [Bpad][rows]) -- given (non-null) when the sample-major kernel is to be used, in which case dyT / inT may be null.
static int run_wgrad(dposer_scorefc_s* h, const void* dyT, int n_rows_pad, int n_valid, const void* inT, int k_rows_pad, int k_valid,
                     int64_t Bpad, float* slabs, int64_t& slab_cursor, int64_t numel, int64_t flat_off, ReduceJobs& rj, hipStream_t st,
                     const void* dy = nullptr, const void* in = nullptr, const Planes* dy_pl = nullptr, const Planes* in_pl = nullptr) {
    const int shape = wgrad_shape(n_rows_pad, k_rows_pad, Bpad);
    const int ct = shape_ct(shape), stt = shape_st(shape);
    const int kb_total = (int)(Bpad / h->KBS);
    const int64_t stages = kb_total / 4;
    const int n_cblk = n_rows_pad / (ct * 32), n_sblk = k_rows_pad / (stt * 32);
    const int ks = pick_ksplit((int64_t)n_cblk * n_sblk, stages, shape == SHAPE_BIG ? 256 : 512);   // resident workgroups of the tiling
    g_next_flops = 2.0 * (double)g_alg_batch * n_valid * k_valid;
    GemmArgs g = gemm_args(dyT, kb_total, n_cblk, n_sblk);
    add_seg(g, inT, kb_total);
    g.ksplit = ks;
    WgradParams p;
    p.slab = slabs + slab_cursor;
    p.slab_stride = numel;
    p.ld = k_valid;
    p.N_valid = n_valid;
    p.K_valid = k_valid;
    if (dy_pl && in_pl) {
        // bf16x3: dW = dy_hi^T in_hi + dy_lo^T in_hi + dy_hi^T in_lo -- three sample-major launches on the bf16 planes into consecutive slab
        // sets, summed by the bucket's reduction like the k-splits of one launch
        for (int term = 0; term < 3; ++term) {
            WgradTrArgs t;
            std::memset(&t, 0, sizeof(t));
            t.dY = term == 1 ? dy_pl->lo : dy_pl->hi; t.H = term == 2 ? in_pl->lo : in_pl->hi;
            t.N = n_rows_pad; t.Kc = k_rows_pad; t.n_cblk = n_cblk; t.n_sblk = n_sblk; t.sblocks = (int)(Bpad / 32); t.ksplit = ks;
            t.alg_flops = term == 0 ? g.alg_flops : 0.0;
            WgradParams pt = p;
            pt.slab = p.slab + (int64_t)term * ks * numel;
            DP_HIP_LAUNCH(gemm_wgrad_tr(shape, t, pt, st));
        }
        ReduceJob& j = rj.job[rj.n++];
        j.dst_off = flat_off; j.count = numel; j.src_off = slab_cursor; j.src_stride = numel; j.nsrc = 3 * ks;
        slab_cursor += (int64_t)3 * ks * numel;
        return DPOSER_OK;
    } else if (dy && in) {
        if (h->f32) return dposer_set_error(DPOSER_ERR_BAD_ARG, "run_wgrad: sample-major operands are bf16 only");
        WgradTrArgs t;
        std::memset(&t, 0, sizeof(t));
        t.dY = dy; t.H = in; t.N = n_rows_pad; t.Kc = k_rows_pad; t.n_cblk = n_cblk; t.n_sblk = n_sblk; t.sblocks = (int)(Bpad / 32); t.ksplit = ks;
        t.alg_flops = g.alg_flops;
        DP_HIP_LAUNCH(gemm_wgrad_tr(shape, t, p, st));
    } else {
        DP_HIP_LAUNCH(gemm_wgrad(h->w32 ? PREC_FP32 : PREC_BF16, shape, g, p, st));
    }
    ReduceJob& j = rj.job[rj.n++];
    j.dst_off = flat_off; j.count = numel; j.src_off = slab_cursor; j.src_stride = numel; j.nsrc = ks;
    slab_cursor += (int64_t)ks * numel;
    return DPOSER_OK;
}

// forward in training layout: every layer input / normalised activation is kept for the backward pass
static int forward_core_train(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, int64_t B, bool dropout_on, uint64_t seed,
                              uint32_t step, hipStream_t st, bool with_post = true) {
    const int L = h->L;
    const bool tr = wgrad_tr_mode(h, w.Bpad);   // then no wgrad reads a transposed activation
    DP_TRY(run_temb(h, flat, packed, w, w.upre, tr ? nullptr : w.tembT, true, st));
    DP_TRY(split_act(h, w.xin, w.p_xin, w.Bpad, h->Dpad, st));
    for (int l = 0; l < L; ++l) {
        const void* in = l == 0 ? (const void*)w.xin : (const void*)w.hbuf[l - 1];
        const Planes& in_pl = l == 0 ? w.p_xin : w.p_h[l - 1];
        const void* resid = (l >= 2 && (l % 2) == 0) ? w.hbuf[l - 2] : nullptr;
        // TRAIN epilogue keeps xhat and the GnAux records (rstd, dropout decisions); dropout only when the module is in train() mode
        const LayerOff& lo = h->layer[l];
        const int shape = gn_shape(w.Bpad, h->H, h->gs, !h->f32);
        const int kx = lo.kin_pad / h->KBS, ke = h->E / h->KBS;
        g_next_flops = 2.0 * (double)B * h->H * (lo.kin + h->E);
        GemmArgs g = gemm_args(packed + h->pk_wl[l], (kx + ke) * h->wk, h->H / (shape_ct(shape) * 32), (int)(w.Bpad / (shape_st(shape) * 32)));
        add_act(h, g, in, in_pl, kx);
        add_act(h, g, w.temb, w.p_temb, ke);
        GNParams p;
        p.bias = reinterpret_cast<const float*>(packed + h->pk_bias_cat) + (int64_t)l * h->H;
        p.gamma = flat + lo.gamma; p.beta = flat + lo.beta; p.out = w.hbuf[l]; p.resid = resid; p.xhat = w.xhat[l]; p.aux = w.aux[l];
        p.H = h->H; p.drop = drop_cfg(h, dropout_on, l, seed, step); p.act = h->d.activation;
        p.outT = tr ? nullptr : w.hT[l]; p.Spad = w.Bpad;     // transposed copy for the wgrad GEMMs of the consuming layer
        p.out_hi = h->x3 ? w.p_h[l].hi : nullptr;             // bf16x3: operand planes from the epilogue; fp32 tiles only where they are a residual input
        p.out_lo = h->x3 ? w.p_h[l].lo : nullptr;
        if (h->x3 && !((l % 2) == 0 && l + 2 < L)) p.out = nullptr;
        DP_HIP_LAUNCH(gemm_gn(gemm_prec(h), true, shape, g, p, st, h->gs));
    }
    if (!with_post) return DPOSER_OK;      // (the fused DSM step runs post_dense itself, with the loss in its epilogue)
    return run_post(h, flat, packed, w.hbuf[L - 1], w.p_h[L - 1], w.res, B, w.Bpad, st);
}

// Who learns that gradient buckets are final, and how (data parallel: the all-reduce of a bucket overlaps with the rest of the backward):
//   events only    : one event per bucket is recorded on the stream (dposer_dsm_loss_fwd_bwd_bucketed); the caller enqueues its waits and
//                    collectives after the call has returned;
//   events + notify: ONE event per group of buckets that become final together, then `notify` is called at once, from inside the call,
//                    with the merged flat ranges -- the host-side cost of issuing the collectives (tens of microseconds each through
//                    torch.distributed) then overlaps with the GPU work still queued, instead of following the whole backward.
struct BucketSink {
    void* const* events = nullptr;
    int n_events = 0;
    dposer_ranges_final_fn notify = nullptr;
    void* user = nullptr;
};
static void bucket_range(const dposer_scorefc_s* h, int b, int64_t& lo, int64_t& hi);
// buckets [first, first + n) of flat_grad are final on `st`
static int mark_final(const dposer_scorefc_s* h, const BucketSink* sink, int first, int n, hipStream_t st) {
    if (!sink || !sink->events || n <= 0) return DPOSER_OK;
    if (!sink->notify) {
        for (int b = first; b < first + n; ++b)
            if (b < sink->n_events && sink->events[b]) DP_CHECK_HIP(hipEventRecord((hipEvent_t)sink->events[b], st));
        return DPOSER_OK;
    }
    if (first >= sink->n_events || !sink->events[first]) return dposer_set_error(DPOSER_ERR_BAD_ARG, "notify: missing bucket event");
    DP_CHECK_HIP(hipEventRecord((hipEvent_t)sink->events[first], st));
    int64_t lo[MAX_L + 1], hi[MAX_L + 1];
    int nr = 0;
    for (int b = first; b < first + n; ++b) {          // buckets come in descending flat order: merge neighbours
        int64_t l, r;
        bucket_range(h, b, l, r);
        if (nr > 0 && lo[nr - 1] == r) lo[nr - 1] = l;
        else { lo[nr] = l; hi[nr] = r; ++nr; }
    }
    sink->notify(sink->user, first, nr, lo, hi, sink->events[first]);
    return DPOSER_OK;
}

// Gradient buckets in the order the backward pass finishes them (dposer_scorefc_grad_buckets), L + 1 of them: bucket b < L-1 is GN
// layer L-1-b (bucket 0 also holds post_dense, which follows the last layer in parameters() order); bucket L-1 ("front A") is layer
// 0's weights [pre_dense.w .. dense_t bias], final as soon as layer 0 is differentiated; bucket L ("front B") is the rest in front of
// layer 1 -- layer 0's GroupNorm affine, gauss_proj.W, the shared time embedding -- final only after the time branch.  The dead
// pre_dense_cond range between the two (1.05 M floats = 13 % of the flat buffer, always zero) belongs to no bucket: it is not
// all-reduced.
static int n_grad_buckets(const dposer_scorefc_s* h) { return h->L + 1; }
static void bucket_range(const dposer_scorefc_s* h, int b, int64_t& lo, int64_t& hi) {
    const int L = h->L;
    if (b < L - 1) {
        const int j = L - 1 - b;
        lo = h->layer[j].w;
        hi = (b == 0) ? h->nparams : h->layer[j + 1].w;
    } else if (b == L - 1) {
        lo = 0;
        hi = h->off_cond_w;
    } else {
        lo = h->layer[0].gamma;
        hi = h->layer[1].w;
    }
}

extern "C" int32_t dposer_scorefc_grad_buckets(dposer_scorefc_t h, int64_t* lo, int64_t* hi, int32_t max_buckets) {
    if (!h) return -1;
    const int n = n_grad_buckets(h);
    for (int b = 0; b < n && b < max_buckets; ++b)
        if (lo && hi) bucket_range(h, b, lo[b], hi[b]);
    return n;
}

extern "C" int dposer_event_create(void** event) {
    DP_CHECK_ARG(event, "null argument");
    hipEvent_t e;
    DP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return DPOSER_OK;
}
extern "C" void dposer_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}
extern "C" int dposer_stream_wait_event(void* stream, void* event) {
    DP_CHECK_ARG(event, "null event");
    DP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return DPOSER_OK;
}

// Should the parameter-gradient half of the backward pass run on the handle's second stream?
// Below ~2 tile rounds per launch (Bpad <= 16384: the per-GPU shard of B = 65536 at 4 and 8 GPUs) single kernels leave CUs idle
// and launch gaps show, so the two wgrad GEMMs + bucket reduction of layer j run concurrently with the dgrad GEMM of layer j-1.
// DPOSER_WGRAD_STREAM = 0 / 1 forces it off / on.
static bool use_side_stream(int64_t Bpad) {
    const int forced = score_tuning().wgrad_stream;
    if (forced >= 0) return forced == 1;
    return Bpad >= 8192 && Bpad <= 16384;     // (re-measured: 1.28 vs 1.36 ms at 16384, a tie at 8192 and 32768, 0.759 vs 0.747 ms at 4096)
}
static int ensure_side_stream(dposer_scorefc_s* h) {
    if (h->side) return DPOSER_OK;
    DP_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    DP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
    DP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_time, hipEventDisableTiming));
    DP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (int l = 0; l < MAX_L; ++l) DP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_layer[l], hipEventDisableTiming));
    return DPOSER_OK;
}

// One launch for every 256 x 256 weight-gradient tile of the step (wgrad_batch.h) instead of two split-K launches per layer: used when
// no gradient bucket has to be final early (single GPU: bucket events are what the data-parallel all-reduce overlaps with), bf16, the
// shipped widths (H = 1024: 4 x 4 tiles, E = 512: 4 x 2 tiles).  Measured against the per-layer launches (with their second stream
// where that was the default), ms per step: 0.489 -> 0.432 at 1280 samples, 0.708 -> 0.561 at 4096, 0.831 -> 0.710 at 8192,
// 1.195 -> 1.069 at 16384, 1.965 -> 1.755 at 32768, 3.41-3.45 -> 3.23 at 65536.
// DPOSER_WGRAD_BATCHED = 0 forces it off, 1 also takes it when bucket events were asked for (they are then all recorded at the end).
// Returns false when not applicable.
// the lane problems of layers [l_first, l_last] (+ the shared time embedding): the whole step, or one layer of the bucketed backward
// term (bf16x3 only): 0 = dy_hi^T in_hi, 1 = dy_lo^T in_hi, 2 = dy_hi^T in_lo on the bf16 planes; -1 = the stored (bf16) arrays
static bool plan_wgrad_lanes(const dposer_scorefc_s* h, const Ws& w, int l_first, int l_last, bool with_se, WgradBatchArgs& a, int term = -1) {
    if (h->H != 1024 || h->E != 512 || h->L < 2 || h->L > 9) return false;
    const int64_t set = (int64_t)WGB_BLOCKS * WGB_MAX_SEG * 65536, need = set * h->wk, small = ((int64_t)14 << 20) * h->wk;
    if (w.slab_elems < need + small) return false;
    if (h->x3 && term < 0) term = 0;
    auto A = [&](const void* stored, const Planes& pl) -> const void* { return term < 0 ? stored : (term == 1 ? pl.lo : pl.hi); };      // dY side
    auto Bs = [&](const void* stored, const Planes& pl) -> const void* { return term < 0 ? stored : (term == 2 ? pl.lo : pl.hi); };     // input side
    const int S = (int)(w.Bpad / 32), nAh = h->H / 16, nE = h->E / 16;
    const int L = l_last + 1;
    if (S % 2 != 0) return false;
    std::memset(&a, 0, sizeof(a));
    int n = 0;
    for (int l = l_first > 1 ? l_first : 1; l < L; ++l) {           // W_x (layer 0 has 63 input channels: its own small kernel)
        if (h->layer[l].kin != h->H || h->layer[l].kin_pad != h->H) return false;
        if (n >= WGB_MAX_PROB) return false;                        // (three-block models: more lane problems than one launch holds)
        WgradLaneProblem& p = a.prob[n++];
        for (int i = 0; i < 2; ++i) {
            p.dY[i] = A(w.dy[l], w.p_dy[l]); p.H[i] = Bs(w.hbuf[l - 1], w.p_h[l - 1]); p.nA[i] = nAh; p.nB[i] = nAh; p.sblk0[i] = 2 * i; p.sb_off[i] = 0;
            p.dst_off[i] = h->layer[l].w; p.ld[i] = h->H;
        }
        p.len = S;
    }
    int lt = l_first;
    for (; lt + 1 < L; lt += 2) {                                   // W_t of two layers side by side
        const int l = lt;
        if (n >= WGB_MAX_PROB) return false;                        // (three-block models: more lane problems than one launch holds)
        WgradLaneProblem& p = a.prob[n++];
        for (int i = 0; i < 2; ++i) {
            p.dY[i] = A(w.dy[l + i], w.p_dy[l + i]); p.H[i] = Bs(w.temb, w.p_temb); p.nA[i] = nAh; p.nB[i] = nE; p.sblk0[i] = 0; p.sb_off[i] = 0;
            p.dst_off[i] = h->layer[l + i].wt; p.ld[i] = h->E;
        }
        p.len = S;
    }
    if (lt < L) {                                                   // the left-over W_t: its rows in two halves
        if (n >= WGB_MAX_PROB) return false;                        // (three-block models: more lane problems than one launch holds)
        WgradLaneProblem& p = a.prob[n++];
        for (int i = 0; i < 2; ++i) {
            p.dY[i] = A(w.dy[L - 1], w.p_dy[L - 1]); p.H[i] = Bs(w.temb, w.p_temb); p.nA[i] = nAh; p.nB[i] = nE; p.sblk0[i] = 0; p.sb_off[i] = i * (S / 2);
            p.dst_off[i] = h->layer[L - 1].wt; p.ld[i] = h->E;
        }
        p.len = S / 2;
        p.split_k = 1;
    }
    if (with_se && S % 4 == 0 && n < WGB_MAX_PROB) {                           // shared time embedding [E x E] = 2 x 2 tiles: four row quarters side by side
        if (n >= WGB_MAX_PROB) return false;                        // (three-block models: more lane problems than one launch holds)
        WgradLaneProblem& p = a.prob[n++];
        p.dY[0] = A(w.dU, w.p_dU); p.H[0] = Bs(w.emb, w.p_emb); p.nA[0] = nE; p.nB[0] = nE; p.dst_off[0] = h->off_se_w; p.ld[0] = h->E;
        p.len = S / 4;
        p.mode = 1;
    }
    if (n > WGB_MAX_PROB) return false;
    a.partials = w.slabs + (w.slab_elems - need) + (term > 0 ? term * set : 0);
    a.nterm = h->wk; a.term_stride = set;
    a.span = ((int64_t)S * nAh) << 10;
    if (a.span >= (int64_t)0xfff00000) return false;
    // every lane: at most WGB_MAX_SEG segments (short ones take the kernel's generic prologue / tail path)
    auto fits = [&](int np) {
        int64_t total = 0;
        for (int i = 0; i < np; ++i) total += a.prob[i].len;
        a.nprob = np;
        a.q = (int)ceil_div(total, WGB_LANES);
        for (int lane = 0; lane < WGB_LANES; ++lane) {
            const int64_t lo = (int64_t)lane * a.q, hi = lo + a.q;
            int64_t start = 0;
            int segs = 0;
            for (int i = 0; i < np; ++i) {
                const int64_t end = start + a.prob[i].len, s0 = lo > start ? lo : start, s1 = hi < end ? hi : end;
                if (s0 < s1) ++segs;
                start = end;
            }
            if (segs > WGB_MAX_SEG) return false;
        }
        return true;
    };
    if (fits(n)) return true;
    if (a.prob[n - 1].mode == 1) {                                  // without the shared embedding (it keeps its own launch)
        std::memset(&a.prob[n - 1], 0, sizeof(WgradLaneProblem));
        return fits(n - 1);
    }
    return false;
}
static bool plan_batched_wgrad(const dposer_scorefc_s* h, const Ws& w, bool tr, bool has_events, WgradBatchArgs& a) {
    const int forced = score_tuning().wgrad_batched;
    if (forced == 0 || !tr || (has_events && forced != 1)) return false;
    return plan_wgrad_lanes(h, w, 0, h->L - 1, true, a);
}
// The bucketed backward (data parallel: a bucket's gradient must be final while the layers in front of it are still being
// differentiated, so that its all-reduce overlaps with them) runs the 256 x 256 weight-gradient tiles as lanes too, one launch per
// GROUP of layers: after the dgrad of the group's lowest layer, W_x and W_t of all its layers are one line of lane problems -- 256
// workgroups of equal length instead of two split-K launches per layer with 128 / 64-stage workgroups.  Round 3 had one group per
// layer from 16384 samples per rank (each output tile then has ~11 partial tiles to add: 3.41 -> 3.37 ms at 65536) and split-K launches
// on a second stream below; round 4 (profiles/r04_dp_rank_step.md): TWO groups by default at every batch size -- layers {L-1, L-2}
// (+ post_dense) first, everything else after layer 0, and BEFORE the time branch, whose dgrad + shared-embedding wgrad then run under
// the second group's all-reduce; only "front B" (1 MB) is final at the very end.
// DPOSER_WGRAD_GROUPS = n picks the number of groups (n = L: one per layer), DPOSER_WGRAD_LAYER_LANES = 1 is n = L, = 0 (or
// DPOSER_WGRAD_BATCHED = 0) keeps the split-K launches.
static int plan_wgrad_groups(const dposer_scorefc_s* h, const Ws& w, bool tr, int* lo, int* hi) {
    const ScoreTuning& tn = score_tuning();
    if (tn.wgrad_batched == 0 || tn.wgrad_layer_lanes == 0 || tn.wgrad_groups == 0 || !tr) return 0;
    const int L = h->L;
    int G = tn.wgrad_groups > 0 ? tn.wgrad_groups : (tn.wgrad_layer_lanes == 1 ? L : 2);
    if (G > L) G = L;
    // contiguous groups from the last layer down; the groups at the bottom take the left-over layers ({4,3} {2,1,0} for L = 5, G = 2)
    const int base = L / G, extra = L % G;
    int top = L - 1;
    for (int g = 0; g < G; ++g) {
        const int size = base + (g >= G - extra ? 1 : 0);
        hi[g] = top;
        lo[g] = top - size + 1;
        top -= size;
        WgradBatchArgs probe;
        if (!plan_wgrad_lanes(h, w, lo[g], hi[g], false, probe)) return 0;
    }
    return G;
}

// backward from dres (FT [Bpad][Cp], zero on padded rows) to the flat parameter gradient and/or dx.
//   critical path (stream st): dgrad GEMM of layer L-1 ... 0 (each writes dy_j and the GroupNorm partial sums), dx GEMM, dgrad into the
//                              time branch;
//   gradient side, one of:     (batched)  everything at the end: one lane launch for all 256 x 256 tiles (single GPU);
//                              (grouped)  one lane launch per layer group + that group's reductions and bucket events (data parallel);
//                              (split-K)  per layer two split-K wgrad GEMMs + the bucket's reduction, optionally on the handle's second
//                                         stream (fp32 mode, other widths, forced).
static int backward_core(dposer_scorefc_s* h, const float* flat, const char* packed, Ws& w, int64_t B, bool dropout_on, uint64_t seed,
                         uint32_t step, float* flat_grad, float* dx, const BucketSink* sink, hipStream_t st,
                         const SumJob* loss_sum = nullptr, int dsm_cs_rows = 0) {
    const int prec = gemm_prec(h);
    const int H = h->H, E = h->E, L = h->L, KBS = h->KBS, wk = h->wk;
    const int64_t Bpad = w.Bpad;
    const bool want_w = flat_grad != nullptr;
    const bool tr = wgrad_tr_mode(h, Bpad);
    const bool has_events = sink != nullptr && sink->events != nullptr && sink->n_events > 0;
    WgradBatchArgs wb;
    const bool batched = want_w && plan_batched_wgrad(h, w, tr, has_events, wb);      // (bf16x3: one lane launch per product term, the partial sets added by one reduction)
    int grp_lo[MAX_L], grp_hi[MAX_L];
    const int n_groups = (want_w && !batched && !h->x3) ? plan_wgrad_groups(h, w, tr, grp_lo, grp_hi) : 0;
    // bf16x3: dres arrives from an elementwise kernel as fp32 fragment tiles
    DP_TRY(split_act(h, w.dres, w.p_dres, Bpad, h->Cp, st));
    const bool grouped = n_groups > 0;
    int cur_group = 0;
    bool front_a_done = false;
    const bool two = want_w && !batched && !grouped && use_side_stream(Bpad);
    if (two) DP_TRY(ensure_side_stream(h));
    hipStream_t sw = two ? h->side : st;
    ReduceJobs rj;
    rj.n = 0;
    rj.alt = nullptr;
    int64_t slab_cursor = 0;
    int n_chunks_post = 0, silu_rows = 0;
    const int64_t lane_room = (int64_t)WGB_BLOCKS * WGB_MAX_SEG * 65536 * h->wk;
    // Deterministic reduction into the flat gradient: every partial buffer lives in the workspace (offsets relative to
    // w.slabs); the jobs of one bucket are launched together as soon as its last wgrad has been queued.
    auto rel = [&](const float* p) { return (int64_t)(p - w.slabs); };
    auto add_job = [&](int64_t dst, int64_t count, const float* src, int64_t stride, int nsrc) {
        ReduceJob& jb = rj.job[rj.n++];
        jb.dst_off = dst; jb.count = count; jb.src_off = rel(src); jb.src_stride = stride; jb.nsrc = nsrc;
    };
    if (want_w) {
        if (two) {   // everything the gradient side reads at this point (dres, forward activations) is complete on st
            DP_CHECK_HIP(hipEventRecord(h->ev_start, st));
            DP_CHECK_HIP(hipStreamWaitEvent(sw, h->ev_start, 0));
        }
        // operands that only depend on the forward pass
        if (!tr) {
            DP_HIP_LAUNCH(launch_ft_transpose(h->f32, w.xin, w.xinT, Bpad, h->Dpad, sw));
            DP_HIP_LAUNCH(launch_ft_transpose(h->f32, w.emb, w.embT, Bpad, E, sw));
        }
        // post_dense: bias (column sums of dres: left by k_dsm in the fused step, else a k_colsum launch here) and weight
        if (dsm_cs_rows > 0) n_chunks_post = dsm_cs_rows;
        else DP_HIP_LAUNCH(launch_colsum(h->f32, w.dres, w.cs_part_post, Bpad, h->Cp, &n_chunks_post, sw, nullptr));
        if (h->x3) {
            DP_TRY(run_wgrad(h, nullptr, h->Cp, h->D, nullptr, H, H, Bpad, w.slabs, slab_cursor, (int64_t)h->D * H, h->off_post_w, rj, sw, nullptr, nullptr, &w.p_dres, &w.p_h[L - 1]));
        } else if (tr) {
            DP_TRY(run_wgrad(h, nullptr, h->Cp, h->D, nullptr, H, H, Bpad, w.slabs, slab_cursor, (int64_t)h->D * H, h->off_post_w, rj, sw, w.dres, w.hbuf[L - 1]));
        } else {
            DP_HIP_LAUNCH(launch_ft_transpose(h->f32, w.dres, w.dresT, Bpad, h->Cp, sw));
            DP_TRY(run_wgrad(h, w.dresT, h->Cp, h->D, w.hT[L - 1], H, H, Bpad, w.slabs, slab_cursor, (int64_t)h->D * H, h->off_post_w, rj, sw));
        }
    } else if (loss_sum && loss_sum->n > 0) {
        DP_HIP_LAUNCH(launch_sum_partials(loss_sum->part, loss_sum->n, loss_sum->out, st));
    }
    const int gshape = gnbwd_shape(Bpad, h->gs, !h->f32);
    const int ws_rows = (int)(Bpad / (shape_st(gshape) * 32)) * shape_ws(gshape);   // partial rows written by the dgrad epilogue
    // GroupNorm affine / bias gradients of layer j from the partial sums its dgrad epilogue wrote
    auto add_layer_jobs = [&](int j) {
        const LayerOff& lo = h->layer[j];
        add_job(lo.gamma, H, w.gn_part[j] + 0 * H, 3 * (int64_t)H, ws_rows);
        add_job(lo.beta, H, w.gn_part[j] + 1 * H, 3 * (int64_t)H, ws_rows);
        add_job(lo.b, H, w.gn_part[j] + 2 * H, 3 * (int64_t)H, ws_rows);
        add_job(lo.bt, H, w.gn_part[j] + 2 * H, 3 * (int64_t)H, ws_rows);
        if (j == L - 1) add_job(h->off_post_b, h->D, w.cs_part_post, h->Cp, n_chunks_post);
    };
    auto run_wx0 = [&]() -> int {      // layer 0's W_x has 63 input channels: not a 256-wide lane problem, its own small split-K launch
        const LayerOff& lo = h->layer[0];
        if (h->x3) return run_wgrad(h, nullptr, H, H, nullptr, lo.kin_pad, lo.kin, Bpad, w.slabs, slab_cursor, (int64_t)H * lo.kin, lo.w, rj, sw, nullptr, nullptr, &w.p_dy[0], &w.p_xin);
        return run_wgrad(h, nullptr, H, H, nullptr, lo.kin_pad, lo.kin, Bpad, w.slabs, slab_cursor, (int64_t)H * lo.kin, lo.w, rj, sw, w.dy[0], (const void*)w.xin);
    };
    for (int j = L - 1; j >= 0; --j) {
        // gradient w.r.t. the output of GN layer j, through the layer that consumes it
        const bool from_post = (j == L - 1);
        const void* Wt = packed + (from_post ? h->pk_wpostT : h->pk_wlT[j + 1]);
        const int kblocks = (from_post ? h->Cp : H) / KBS;
        g_next_flops = 2.0 * (double)B * H * (from_post ? h->D : H);
        GemmArgs g = gemm_args(Wt, kblocks * wk, H / (shape_ct(gshape) * 32), (int)(Bpad / (shape_st(gshape) * 32)));
        add_act(h, g, from_post ? (const void*)w.dres : (const void*)w.dy[j + 1], from_post ? w.p_dres : w.p_dy[j + 1], kblocks);
        GNBwdParams p;
        const bool even = (j % 2) == 0;
        p.carry_in = (even && j < L - 1) ? w.carry[(j / 2 + 1) & 1] : nullptr;
        p.carry_out = (even && j >= 2) ? w.carry[(j / 2) & 1] : nullptr;
        p.xhat = w.xhat[j]; p.aux = w.aux[j]; p.gamma = flat + h->layer[j].gamma; p.beta = flat + h->layer[j].beta;
        p.dy = w.dy[j]; p.part = w.gn_part[j]; p.H = H; p.S_valid = B;
        p.drop_scale = (dropout_on && h->d.dropout_p > 0.f) ? 1.0f / (1.0f - h->d.dropout_p) : 1.0f;   // the decisions themselves come from the forward pass (GnAux)
        p.dyT = (want_w && !tr) ? w.dyT[j] : nullptr; p.Spad = Bpad; p.act = h->d.activation;
        p.dy_hi = h->x3 ? w.p_dy[j].hi : nullptr;            // bf16x3: dy is only ever a GEMM operand -- its planes, no fp32 tiles
        p.dy_lo = h->x3 ? w.p_dy[j].lo : nullptr;
        if (h->x3) p.dy = nullptr;
        DP_HIP_LAUNCH(gemm_gn_bwd(prec, gshape, g, p, st, h->gs));
        if (!want_w) continue;
        if (two) {
            DP_CHECK_HIP(hipEventRecord(h->ev_layer[j], st));
            DP_CHECK_HIP(hipStreamWaitEvent(sw, h->ev_layer[j], 0));
        }
        // parameter gradients of layer j
        const LayerOff& lo = h->layer[j];
        if (batched) {       // W_x (j >= 1) and W_t of every layer are tiles of the one launch behind the loop
            if (j == 0) DP_TRY(run_wx0());
            add_layer_jobs(j);
        } else if (grouped) {
            if (j != grp_lo[cur_group]) continue;
            // the group [grp_lo, grp_hi] is differentiated: its lanes, its partial tiles, its small reductions, its bucket events
            if (j == 0) DP_TRY(run_wx0());
            if (slab_cursor > w.slab_elems - lane_room) return dposer_set_error(DPOSER_ERR_BAD_ARG, "backward: slab buffer too small for the lane launch");
            if (!plan_wgrad_lanes(h, w, grp_lo[cur_group], grp_hi[cur_group], false, wb)) return dposer_set_error(DPOSER_ERR_BAD_ARG, "backward: lane plan changed between probe and launch");
            double fl = 0.0;
            for (int l = grp_lo[cur_group]; l <= grp_hi[cur_group]; ++l) fl += 2.0 * (double)B * H * ((l >= 1 ? (double)H : 0.0) + (double)E);
            wb.alg_flops = fl;
            DP_HIP_LAUNCH(gemm_wgrad_tr_batch(wb, sw));
            for (int l = grp_hi[cur_group]; l >= grp_lo[cur_group]; --l) add_layer_jobs(l);
            DP_HIP_LAUNCH(launch_reduce_all(wb, rj, w.slabs, flat_grad, sw));          // partial tiles + the group's small reductions: one launch
            rj.n = 0;
            DP_TRY(mark_final(h, sink, L - 1 - grp_hi[cur_group], grp_hi[cur_group] - grp_lo[cur_group] + 1, sw));
            if (j == 0) front_a_done = true;
            ++cur_group;
        } else {
            const void* inT = (j == 0) ? (const void*)w.xinT : (const void*)w.hT[j - 1];
            if (h->x3) {
                DP_TRY(run_wgrad(h, nullptr, H, H, nullptr, lo.kin_pad, lo.kin, Bpad, w.slabs, slab_cursor, (int64_t)H * lo.kin, lo.w, rj, sw, nullptr, nullptr, &w.p_dy[j], j == 0 ? &w.p_xin : &w.p_h[j - 1]));
                DP_TRY(run_wgrad(h, nullptr, H, H, nullptr, E, E, Bpad, w.slabs, slab_cursor, (int64_t)H * E, lo.wt, rj, sw, nullptr, nullptr, &w.p_dy[j], &w.p_temb));
            } else {
            if (tr) DP_TRY(run_wgrad(h, nullptr, H, H, nullptr, lo.kin_pad, lo.kin, Bpad, w.slabs, slab_cursor, (int64_t)H * lo.kin, lo.w, rj, sw, w.dy[j], j == 0 ? (const void*)w.xin : (const void*)w.hbuf[j - 1]));
            else DP_TRY(run_wgrad(h, w.dyT[j], H, H, inT, lo.kin_pad, lo.kin, Bpad, w.slabs, slab_cursor, (int64_t)H * lo.kin, lo.w, rj, sw));
            if (tr) DP_TRY(run_wgrad(h, nullptr, H, H, nullptr, E, E, Bpad, w.slabs, slab_cursor, (int64_t)H * E, lo.wt, rj, sw, w.dy[j], w.temb));
            else DP_TRY(run_wgrad(h, w.dyT[j], H, H, w.tembT, E, E, Bpad, w.slabs, slab_cursor, (int64_t)H * E, lo.wt, rj, sw));
            }
            add_layer_jobs(j);
            // layer j's gradient (and everything behind it in the flat buffer) is final: the data-parallel all-reduce of this
            // bucket can start while the remaining layers are still being differentiated
            if (rj.n > 0) DP_HIP_LAUNCH(launch_reduce_grads(rj, w.slabs, flat_grad, sw));
            rj.n = 0;
            DP_TRY(mark_final(h, sink, L - 1 - j, 1, sw));
            if (j == 0) front_a_done = true;
        }
    }
    if (dx) {   // d loss / d x = dy_0 @ W_pre  (the time branch does not depend on x)
        const int shape = final_shape(Bpad);
        g_next_flops = 2.0 * (double)B * H * h->D;
        GemmArgs g = gemm_args(packed + h->pk_wlT[0], H / KBS * wk, h->Dpad / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
        add_act(h, g, w.dy[0], w.p_dy[0], H / KBS);
        RowMajorParams p;
        p.bias = nullptr; p.out = dx; p.ldc = h->D; p.C_valid = h->D; p.S_valid = B;
        DP_HIP_LAUNCH(gemm_rowmajor(prec, shape, g, p, st));
    }
    if (!want_w) return DPOSER_OK;
    // time branch: dtemb = sum_l dy_l @ Wt_l ; dU = dtemb * silu'(u)
    {
        // (at 8192 samples this is 256 workgroups of 128x128 with K = L*H = 5120 -- one per CU, latency-bound, 121 us; the
        //  128x32 tiling with 4x the workgroups was measured slower, 144 us)
        const int shape = main_shape(Bpad, E);
        if (h->x3) {
            // bf16x3: one GEMM per layer (three plane segments against that layer's [E][3 H] matrix) into fp32 partials; k_silu_bwd_reduce adds
            // them in layer order, applies act'(u) and keeps the column sums -- the small-batch form of the bf16 mode, at every batch size
            for (int l = 0; l < L; ++l) {
                g_next_flops = 2.0 * (double)B * E * H;
                GemmArgs gl = gemm_args(packed + h->pk_wtT_all + (int64_t)l * E * H * 2 * wk, H / KBS * wk, E / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
                add_act(h, gl, w.dy[l], w.p_dy[l], H / KBS);
                PartialFTParams pp;
                pp.out = w.dU_part + (int64_t)l * Bpad * E; pp.N = E; pp.split_stride = 0;
                DP_HIP_LAUNCH(gemm_partial_ft(prec, shape, gl, pp, st));
            }
            SiLUBwdReduceArgs ra;
            ra.part = w.dU_part; ra.nsplit = L; ra.split_stride = Bpad * (int64_t)E; ra.pre = w.upre; ra.out = w.dU; ra.cs_part = w.silu_part;
            ra.N = E; ra.act = h->d.activation; ra.f32 = h->f32; ra.B = B; ra.Spad = Bpad;
            int nb = 0;
            DP_HIP_LAUNCH(launch_silu_bwd_reduce_ft(ra, w.p_dU.hi, w.p_dU.lo, (int)silu_part_rows(Bpad, true), &nb, st));      // (writes dU's operand planes too)
            silu_rows = nb;
        } else {
        g_next_flops = 2.0 * (double)B * E * L * H;
        GemmArgs g = gemm_args(packed + h->pk_wtT_all, L * H / KBS, E / (shape_ct(shape) * 32), (int)(Bpad / (shape_st(shape) * 32)));
        for (int l = 0; l < L; ++l) add_seg(g, w.dy[l], H / KBS);
        // Small batches: 40 tiles at 1280 samples, each walking K = L * H alone (36 us).  One k-split per layer segment fills the chip
        // (12 us for the plain GEMM at 1280 samples, tools/tune_gemm.hip TUNE_SPLITK); k_silu_bwd_reduce adds the splits in layer order,
        // applies act'(u) and keeps the column sums.  From ~4096 samples up the one launch is as fast: DPOSER_SILU_SPLIT_MAX.
        if (tr && (shape == SHAPE_MID || shape == SHAPE_SMALL) && w.dU_part && Bpad <= score_tuning().silu_split_max && Bpad <= SILU_SPLIT_CAP && L <= GEMM_MAX_SEG && L <= 8) {
            g.ksplit = L;
            g.split_segments = 1;
            PartialFTParams pp;
            pp.out = w.dU_part; pp.N = E; pp.split_stride = Bpad * (int64_t)E;
            DP_HIP_LAUNCH(gemm_partial_ft(prec, shape, g, pp, st));
            SiLUBwdReduceArgs ra;
            ra.part = w.dU_part; ra.nsplit = L; ra.split_stride = Bpad * (int64_t)E; ra.pre = w.upre; ra.out = w.dU; ra.cs_part = w.silu_part;
            ra.N = E; ra.act = h->d.activation; ra.f32 = h->f32; ra.B = B; ra.Spad = Bpad;
            int nb = 0;
            DP_HIP_LAUNCH(launch_silu_bwd_reduce(ra, (int)silu_part_rows(Bpad), &nb, st));
            silu_rows = nb;
        } else {
        SiLUBwdParams p;
        p.pre = w.upre; p.out = w.dU; p.N = E; p.S_valid = B; p.outT = tr ? nullptr : w.dUT; p.Spad = Bpad; p.act = h->d.activation;
        p.part = w.silu_part;                                            // the shared embedding's bias gradient: no column-sum launch over dU
        std::memset(&p.drop, 0, sizeof(p.drop));
        silu_rows = (int)(Bpad / (shape_st(shape) * 32)) * shape_ws(shape);
        DP_HIP_LAUNCH(gemm_silu_bwd(prec, shape, g, p, st));
        }
        }
    }
    if (two) {
        DP_CHECK_HIP(hipEventRecord(h->ev_time, st));
        DP_CHECK_HIP(hipStreamWaitEvent(sw, h->ev_time, 0));
    }
    const bool se_in_batch = batched && wb.prob[wb.nprob - 1].mode == 1;
    if (!se_in_batch) {                                             // (otherwise: a lane problem of the one launch below)
        if (h->x3) DP_TRY(run_wgrad(h, nullptr, E, E, nullptr, E, E, Bpad, w.slabs, slab_cursor, (int64_t)E * E, h->off_se_w, rj, sw, nullptr, nullptr, &w.p_dU, &w.p_emb));
        else if (tr) DP_TRY(run_wgrad(h, nullptr, E, E, nullptr, E, E, Bpad, w.slabs, slab_cursor, (int64_t)E * E, h->off_se_w, rj, sw, w.dU, w.emb));
        else DP_TRY(run_wgrad(h, w.dUT, E, E, w.embT, E, E, Bpad, w.slabs, slab_cursor, (int64_t)E * E, h->off_se_w, rj, sw));
    }
    if (batched) {
        if (slab_cursor > w.slab_elems - lane_room) return dposer_set_error(DPOSER_ERR_BAD_ARG, "backward: slab buffer too small for the batched wgrad launch");
        wb.alg_flops = 2.0 * (double)B * H * ((double)(L - 1) * H + (double)L * E) + (se_in_batch ? 2.0 * (double)B * E * E : 0.0);
        DP_HIP_LAUNCH(gemm_wgrad_tr_batch(wb, st));
        if (h->x3) {     // the other two product terms: the same plan on the other planes, each into its own partial set
            for (int term = 1; term < 3; ++term) {
                WgradBatchArgs wt;
                if (!plan_wgrad_lanes(h, w, 0, L - 1, se_in_batch, wt, term) || wt.nprob != wb.nprob) return dposer_set_error(DPOSER_ERR_BAD_ARG, "backward: lane plan changed between the product terms");
                wt.alg_flops = 0.0;
                DP_HIP_LAUNCH(gemm_wgrad_tr_batch(wt, st));
            }
        }
    }
    // what is left: the shared time embedding (front B), layer 0's jobs where they were not flushed with a group, and the parameters
    // that never get a gradient
    add_job(h->off_se_b, E, w.silu_part, E, silu_rows);
    if (loss_sum && loss_sum->n > 0) {                              // the step's loss value: summed by one block of the last reduction
        ReduceJob& jb = rj.job[rj.n++];
        jb.dst_off = 0; jb.count = loss_sum->n; jb.src_off = rel(loss_sum->part); jb.src_stride = 0; jb.nsrc = -1;
        rj.alt = loss_sum->out;
    }
    for (int i = 0; i < h->n_nograd; ++i) {                          // dead parameters: zeros (no bucket holds them; the optimiser skips them)
        ReduceJob& jb = rj.job[rj.n++];
        jb.dst_off = h->nograd_lo[i]; jb.count = h->nograd_hi[i] - h->nograd_lo[i]; jb.src_off = 0; jb.src_stride = 0; jb.nsrc = 0;
    }
    if (batched) DP_HIP_LAUNCH(launch_reduce_all(wb, rj, w.slabs, flat_grad, st));      // (batched: sw == st) partial tiles + every small reduction of the step
    else if (rj.n > 0) DP_HIP_LAUNCH(launch_reduce_grads(rj, w.slabs, flat_grad, sw));
    rj.n = 0;
    if (batched && has_events) {     // nothing was final before this point: every bucket becomes final with the last reduction
        DP_TRY(mark_final(h, sink, 0, L + 1, sw));
    } else {
        if (!front_a_done) DP_TRY(mark_final(h, sink, L - 1, 2, sw));      // front A + front B
        else DP_TRY(mark_final(h, sink, L, 1, sw));                         // front B (front A: recorded with layer 0 / its group)
    }
    if (two) {   // the caller's stream owns the complete gradient (and may reuse the workspace) from here on
        DP_CHECK_HIP(hipEventRecord(h->ev_join, sw));
        DP_CHECK_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    }
    return DPOSER_OK;
}

static int dsm_loss_fwd_bwd_impl(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const dposer_sde_desc* sde,
                                 const float* batch_x, const float* t_in, const float* z_in, float eps, uint64_t seed, uint32_t step,
                                 const float* freq, const float* sigmas, float* flat_grad, float* loss, int64_t B, const BucketSink* sink,
                                 void* stream) {
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(sde && batch_x && freq && sigmas && flat_grad && loss, "null argument");
    DP_CHECK_ARG(sde_kind_ok(sde), "unknown SDE kind");
    DP_CHECK_ARG(sde->kind != DPOSER_SDE_VE_DISCRETE && sde->kind != DPOSER_SDE_VP_DISCRETE,
                 "the denoising-score-matching step takes the continuous score function (discrete models train on the SMLD / DDPM losses)");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_TRAIN, 0, (char*)ws_, w);
    const int64_t Bpad = w.Bpad;
    const SdeCfg sc = to_sde(sde);
    PrepTrainArgs pa;
    pa.x0 = batch_x; pa.t_in = t_in; pa.z_in = z_in; pa.freq = freq; pa.xin = w.xin; pa.emb = w.emb; pa.t_out = w.tbuf; pa.z_out = w.zbuf;
    pa.B = B; pa.Bpad = Bpad; pa.D = h->D; pa.Dpad = h->Dpad; pa.E = h->E; pa.fourier = h->d.embedding == DPOSER_EMB_FOURIER; pa.f32 = h->f32; pa.sde = sc; pa.eps = eps;
    pa.seed = seed; pa.step = step;
    DP_HIP_LAUNCH(launch_prep_train(pa, st));
    // post_dense + loss + d loss / d res: one launch (EpiDsm) while its partials fit the step's buffers, else GEMM -> res -> k_dsm
    const int pshape = final_shape(Bpad);
    const int64_t prow = (Bpad / (shape_st(pshape) * 32)) * shape_ws(pshape);                 // wave rows = column-sum partial rows
    const int chan_waves = pshape == SHAPE_FINAL ? 1 : 2;                                    // waves side by side over the 64 channels
    const bool fused_post = score_tuning().dsm_fused && !h->x3 && sde->kind != DPOSER_SDE_VE && h->Cp == 64 && prow * chan_waves <= 8192 && prow <= cs_post_rows(Bpad);
    DP_TRY(forward_core_train(h, flat, packed, w, B, true, seed, step, st, !fused_post));
    int nb = 0;
    if (fused_post) {
        g_next_flops = 2.0 * (double)B * h->D * h->H;
        GemmArgs g = gemm_args(packed + h->pk_wpost, h->H / h->KBS, h->Cp / (shape_ct(pshape) * 32), (int)(Bpad / (shape_st(pshape) * 32)));
        add_seg(g, w.hbuf[h->L - 1], h->H / h->KBS);
        DsmStepParams p;
        std::memset(&p, 0, sizeof(p));
        p.bias = flat + h->off_post_b; p.t = w.tbuf; p.z = w.zbuf; p.sigmas = sigmas; p.dres = w.dres; p.loss_part = w.loss_part;
        p.cs_part = w.cs_part_post; p.sde = make_sde_dev(sc); p.grad_scale = (float)(1.0 / ((double)B * (double)h->D));
        p.num_scales = h->d.num_scales; p.scale_by_sigma = h->d.scale_by_sigma; p.fourier = h->d.embedding == DPOSER_EMB_FOURIER;
        p.D = h->D; p.Dpad = h->Dpad; p.Cp = h->Cp; p.S_valid = B;
        DP_HIP_LAUNCH(gemm_dsm_step(h->f32 ? PREC_FP32 : PREC_BF16, pshape, g, p, st));
        nb = (int)(prow * chan_waves);
        const SumJob loss_sum{w.loss_part, nb, loss};
        return backward_core(h, flat, packed, w, B, true, seed, step, flat_grad, nullptr, sink, st, &loss_sum, (int)prow);
    }
    DsmArgs da;
    da.res = w.res; da.t = w.tbuf; da.z = w.zbuf; da.sigmas = sigmas; da.dres = w.dres; da.loss_part = w.loss_part; da.B = B; da.Bpad = Bpad;
    da.cs_part = w.cs_part_post;
    da.D = h->D; da.Dpad = h->Dpad; da.Cp = h->Cp; da.num_scales = h->d.num_scales; da.scale_by_sigma = h->d.scale_by_sigma;
    da.f32 = h->f32; da.fourier = h->d.embedding == DPOSER_EMB_FOURIER; da.grad_scale = (float)(1.0 / ((double)B * (double)h->D)); da.sde = sc;
    DP_HIP_LAUNCH(launch_dsm(da, &nb, st));
    const SumJob loss_sum{w.loss_part, nb, loss};                      // summed by one block of the LAST reduction launch of the backward
    return backward_core(h, flat, packed, w, B, true, seed, step, flat_grad, nullptr, sink, st, &loss_sum, nb);
}

extern "C" int dposer_dsm_loss_fwd_bwd_bucketed(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_,
                                                const dposer_sde_desc* sde, const float* batch_x, const float* t_in, const float* z_in,
                                                float eps, uint64_t seed, uint32_t step, const float* freq, const float* sigmas,
                                                float* flat_grad, float* loss, int64_t B, void* const* bucket_events, int32_t n_events,
                                                void* stream) {
    DP_RANGE();
    BucketSink sink;
    sink.events = bucket_events; sink.n_events = n_events;
    return dsm_loss_fwd_bwd_impl(h, flat, packed_, ws_, sde, batch_x, t_in, z_in, eps, seed, step, freq, sigmas, flat_grad, loss, B,
                                 (bucket_events && n_events > 0) ? &sink : nullptr, stream);
}

extern "C" int dposer_dsm_loss_fwd_bwd_notify(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_,
                                              const dposer_sde_desc* sde, const float* batch_x, const float* t_in, const float* z_in,
                                              float eps, uint64_t seed, uint32_t step, const float* freq, const float* sigmas,
                                              float* flat_grad, float* loss, int64_t B, void* const* bucket_events, int32_t n_events,
                                              dposer_ranges_final_fn notify, void* user, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(bucket_events && n_events > 0 && notify, "notify form needs bucket events and a callback");
    DP_CHECK_ARG(h && n_events >= n_grad_buckets(h), "one event per gradient bucket (dposer_scorefc_grad_buckets)");
    BucketSink sink;
    sink.events = bucket_events; sink.n_events = n_events; sink.notify = notify; sink.user = user;
    return dsm_loss_fwd_bwd_impl(h, flat, packed_, ws_, sde, batch_x, t_in, z_in, eps, seed, step, freq, sigmas, flat_grad, loss, B, &sink,
                                 stream);
}

extern "C" int dposer_dsm_loss_fwd_bwd(dposer_scorefc_t h, const float* flat, const void* packed, void* ws, const dposer_sde_desc* sde,
                                       const float* batch_x, const float* t_in, const float* z_in, float eps, uint64_t seed,
                                       uint32_t step, const float* freq, const float* sigmas, float* flat_grad, float* loss, int64_t B,
                                       void* stream) {
    DP_RANGE();
    return dposer_dsm_loss_fwd_bwd_bucketed(h, flat, packed, ws, sde, batch_x, t_in, z_in, eps, seed, step, freq, sigmas, flat_grad, loss, B,
                                            nullptr, 0, stream);
}

// ScoreModelFC.forward with everything kept for autograd (model.py:141-196); dropout when train_mode != 0
extern "C" int dposer_scorefc_forward_train(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const float* x,
                                            const float* labels, const float* freq, const float* sigmas, float* out, int64_t B,
                                            int32_t train_mode, uint64_t seed, uint32_t step, void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(x && labels && freq && sigmas && out, "null tensor argument");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_TRAIN, 0, (char*)ws_, w);
    PrepArgs pa;
    pa.x = x; pa.labels = labels; pa.freq = freq; pa.xin = w.xin; pa.emb = w.emb;
    pa.B = B; pa.Bpad = w.Bpad; pa.D = h->D; pa.Dpad = h->Dpad; pa.E = h->E;
    pa.fourier = h->d.embedding == DPOSER_EMB_FOURIER; pa.f32 = h->f32;
    DP_HIP_LAUNCH(launch_prep_infer(pa, st));
    DP_TRY(forward_core_train(h, flat, packed, w, B, train_mode != 0, seed, step, st));
    OutModelArgs oa;
    oa.res = w.res; oa.labels = labels; oa.sigmas = sigmas; oa.out = out; oa.B = B; oa.D = h->D; oa.Cp = h->Cp;
    oa.num_scales = h->d.num_scales; oa.scale_by_sigma = h->d.scale_by_sigma; oa.fourier = pa.fourier;
    DP_HIP_LAUNCH(launch_out_model(oa, st));
    return DPOSER_OK;
}

// backward of dposer_scorefc_forward_train on the same workspace: d/d params (flat_grad, may be NULL)
// and d/d x (dx [B, D], may be NULL) for an upstream gradient dout [B, D].
extern "C" int dposer_scorefc_backward(dposer_scorefc_t h, const float* flat, const void* packed_, void* ws_, const float* labels,
                                       const float* sigmas, const float* dout, float* flat_grad, float* dx, int64_t B,
                                       int32_t train_mode, uint64_t seed, uint32_t step, void* stream) {
    DP_RANGE();
    DP_TRY(check_common(h, flat, packed_, ws_, B));
    g_alg_batch = B;
    DP_CHECK_ARG(labels && sigmas && dout, "null tensor argument");
    hipStream_t st = (hipStream_t)stream;
    const char* packed = (const char*)packed_;
    Ws w;
    layout_ws(h, B, DPOSER_WS_TRAIN, 0, (char*)ws_, w);
    DresArgs da;
    da.dout = dout; da.labels = labels; da.sigmas = sigmas; da.dres = w.dres; da.B = B; da.Bpad = w.Bpad; da.D = h->D; da.Cp = h->Cp;
    da.num_scales = h->d.num_scales; da.scale_by_sigma = h->d.scale_by_sigma; da.fourier = h->d.embedding == DPOSER_EMB_FOURIER;
    da.f32 = h->f32;
    DP_HIP_LAUNCH(launch_dres_from_dout(da, st));
    return backward_core(h, flat, packed, w, B, train_mode != 0, seed, step, flat_grad, dx, nullptr, st);
}

extern "C" int dposer_grad_sqnorm(const float* grad, int64_t n, float* scratch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(grad && scratch && n >= 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int nb = 0;
    DP_HIP_LAUNCH(launch_sqnorm(grad, n, scratch + 16, &nb, st));
    DP_HIP_LAUNCH(launch_sum_partials(scratch + 16, nb, scratch, st));
    return DPOSER_OK;
}
static int adam_step_impl(float* flat, const float* grad, float* m, float* v, float* ema, int64_t n, const int64_t* skip_lo_host,
                          const int64_t* skip_hi_host, int32_t n_skip, double lr, double beta1, double beta2, double eps, double grad_clip,
                          double grad_scale, int64_t adam_step, double ema_one_minus_decay, float* scratch, bool presummed, void* stream,
                          double weight_decay);
extern "C" int dposer_adam_ema_clip_step(float* flat, const float* grad, float* m, float* v, float* ema, int64_t n,
                                         const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip, double lr,
                                         double beta1, double beta2, double eps, double grad_clip, double grad_scale,
                                         int64_t adam_step, double ema_one_minus_decay, float* scratch, void* stream) {
    DP_RANGE();
    return adam_step_impl(flat, grad, m, v, ema, n, skip_lo_host, skip_hi_host, n_skip, lr, beta1, beta2, eps, grad_clip, grad_scale, adam_step,
                          ema_one_minus_decay, scratch, false, stream, 0.0);
}
extern "C" int dposer_adam_ema_clip_step_wd(float* flat, const float* grad, float* m, float* v, float* ema, int64_t n,
                                            const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip, double lr,
                                            double beta1, double beta2, double eps, double weight_decay, double grad_clip, double grad_scale,
                                            int64_t adam_step, double ema_one_minus_decay, float* scratch, int32_t presummed, void* stream) {
    DP_RANGE();
    return adam_step_impl(flat, grad, m, v, ema, n, skip_lo_host, skip_hi_host, n_skip, lr, beta1, beta2, eps, grad_clip, grad_scale, adam_step,
                          ema_one_minus_decay, scratch, presummed != 0, stream, weight_decay);
}
extern "C" int dposer_adam_ema_clip_step_presummed(float* flat, const float* grad, float* m, float* v, float* ema, int64_t n,
                                                   const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip, double lr,
                                                   double beta1, double beta2, double eps, double grad_clip, double grad_scale,
                                                   int64_t adam_step, double ema_one_minus_decay, float* scratch, void* stream) {
    DP_RANGE();
    return adam_step_impl(flat, grad, m, v, ema, n, skip_lo_host, skip_hi_host, n_skip, lr, beta1, beta2, eps, grad_clip, grad_scale, adam_step,
                          ema_one_minus_decay, scratch, true, stream, 0.0);
}
static int adam_step_impl(float* flat, const float* grad, float* m, float* v, float* ema, int64_t n, const int64_t* skip_lo_host,
                          const int64_t* skip_hi_host, int32_t n_skip, double lr, double beta1, double beta2, double eps, double grad_clip,
                          double grad_scale, int64_t adam_step, double ema_one_minus_decay, float* scratch, bool presummed, void* stream,
                          double weight_decay) {
    DP_CHECK_ARG(flat && grad && m && v && scratch, "null argument");
    DP_CHECK_ARG(adam_step >= 1, "adam_step counts from 1");
    DP_CHECK_ARG(n_skip >= 0 && n_skip <= 2, "at most two no-gradient ranges");
    hipStream_t st = (hipStream_t)stream;
    int nb = 0;
    if (!presummed) DP_HIP_LAUNCH(launch_sqnorm(grad, n, scratch + 16, &nb, st));     // (the partials are added up by k_adam_ema)
    AdamArgs a;
    a.sq_part = scratch + 16; a.n_part = nb;
    a.p = flat; a.g = grad; a.m = m; a.v = v; a.ema = ema; a.n = n;
    for (int i = 0; i < 2; ++i) { a.skip_lo[i] = i < n_skip ? skip_lo_host[i] : 0; a.skip_hi[i] = i < n_skip ? skip_hi_host[i] : 0; }
    a.sqnorm = scratch; a.grad_scale = (float)grad_scale; a.grad_clip = (float)grad_clip;
    const double bc1 = 1.0 - std::pow(beta1, (double)adam_step);
    a.step_size = (float)(lr / bc1);                                   // torch: step_size = lr / bias_correction1
    a.one_minus_beta1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.one_minus_beta2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(beta2, (double)adam_step));
    a.ema_one_minus_decay = (float)ema_one_minus_decay;
    a.weight_decay = (float)weight_decay;
    DP_HIP_LAUNCH(launch_adam_ema(a, st));
    return DPOSER_OK;
}

// Adam + clip + EMA and the re-packing of the weights in ONE pass over the optimizer state (kernels_api.h: AdamPackArgs): the packed
// copies the next forward / backward read are written from the updated values while they are in registers / LDS, instead of being
// re-derived from the flat buffer by dposer_scorefc_pack at the start of the next step (k_pack 31 us + k_bias_cat 5 us per step at
// every batch size).  `packed` must have been produced by dposer_scorefc_pack(with_backward = 1) before (the zero padding of the copies
// is written there, never again).  Bitwise the parameters / moments / EMA of dposer_adam_ema_clip_step_wd and the packed bytes of
// dposer_scorefc_pack.
extern "C" int dposer_scorefc_adam_pack_step(dposer_scorefc_t h, float* flat, const float* grad, float* m, float* v, float* ema, void* packed,
                                             const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip, double lr,
                                             double beta1, double beta2, double eps, double weight_decay, double grad_clip, double grad_scale,
                                             int64_t adam_step, double ema_one_minus_decay, float* scratch, int32_t presummed, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && flat && grad && m && v && packed && scratch, "null argument");
    DP_CHECK_ARG(h->adam_pack.n_tensors > 0, "fused optimizer + re-pack step not available for this configuration");
    DP_CHECK_ARG(adam_step >= 1, "adam_step counts from 1");
    DP_CHECK_ARG(n_skip >= 0 && n_skip <= 2, "at most two no-gradient ranges");
    DP_CHECK_ARG(((uintptr_t)flat & 15) == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)m & 15) == 0 && ((uintptr_t)v & 15) == 0 &&
                     (!ema || ((uintptr_t)ema & 15) == 0) && ((uintptr_t)packed & 255) == 0, "buffers must be 16-byte aligned (packed: 256)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = h->nparams;
    int nb = 0;
    if (!presummed) DP_HIP_LAUNCH(launch_sqnorm(grad, n, scratch + 16, &nb, st));
    AdamPackArgs ap = h->adam_pack;
    AdamArgs& a = ap.a;
    a.sq_part = scratch + 16; a.n_part = nb;
    a.p = flat; a.g = grad; a.m = m; a.v = v; a.ema = ema; a.n = n;
    for (int i = 0; i < 2; ++i) { a.skip_lo[i] = i < n_skip ? skip_lo_host[i] : 0; a.skip_hi[i] = i < n_skip ? skip_hi_host[i] : 0; }
    a.sqnorm = scratch; a.grad_scale = (float)grad_scale; a.grad_clip = (float)grad_clip;
    const double bc1 = 1.0 - std::pow(beta1, (double)adam_step);
    a.step_size = (float)(lr / bc1);
    a.one_minus_beta1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.one_minus_beta2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(beta2, (double)adam_step));
    a.ema_one_minus_decay = (float)ema_one_minus_decay;
    a.weight_decay = (float)weight_decay;
    ap.packed = (unsigned char*)packed;
    ap.write_through = score_tuning().adam_write_through;
    DP_HIP_LAUNCH(launch_adam_pack(ap, st));
    return DPOSER_OK;
}

// ---- profiling hooks ------------------------------------------------------------------------------------
extern "C" void dposer_profile_enable(int32_t on) { gemm_prof_enable(on); }
extern "C" int32_t dposer_profile_num_kinds(void) { return GEMM_PROF_KINDS; }
extern "C" int dposer_profile_collect(double* ms_host, int64_t* launches_host, double* flops_host) {
    DP_CHECK_ARG(ms_host && launches_host && flops_host, "null argument");
    if (gemm_prof_collect(ms_host, reinterpret_cast<long long*>(launches_host), flops_host) != 0)
        return dposer_set_error(DPOSER_ERR_HIP, "dposer_profile_collect: event query failed");
    return DPOSER_OK;
}
extern "C" void dposer_profile_kind_name(int32_t kind, char* out, int32_t n) { gemm_prof_kind_name(kind, out, n); }
