// fd_api.cpp -- host side of libfastdiff_hip.so, the inference boundary (include/fastdiff_hip.h): context, state_dict ingestion (weight-norm
// fold + repack), workspace, the denoiser step sequence, the hipGraph-replayed reverse loop, options.  The rows next to the path and the
// hooks: fd_api_ext.cpp (fastdiff_hip_ext.h); the training operators: fd_api_train.cpp (fastdiff_hip_train.h).
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fd_kernels.h"
#include "fd_host.h"

std::string g_create_error;

// rows of the predictor GEMM's fp16 image per utterance (gx_rows in fd_kernels_kp.hip: 128-frame windows + 2 halo rows)
// (... or, for the Winograd form k_kp_gemm_w, 32 frame PAIRS of four 256-byte sub-rows per 64-frame window: whichever is larger)
static inline int gx_rows_host(int T) { return std::max(((T + 127) / 128) * 128 + 2, ((T + 63) / 64) * 128); }

// ------------------------------------------------------------------------------------------------
// profiling helpers (declared in fd_internal.h)
// ------------------------------------------------------------------------------------------------
bool fd_prof_stamps(const fdk::Launch &L, const char *name, hipEvent_t *e0, hipEvent_t *e1)
{
    fd_context *c = L.ctx;
    if (c->profile != 1 || L.capturing) return false;
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        if (!c->event_pool.empty()) { ev[i] = c->event_pool.back(); c->event_pool.pop_back(); }
        else if (hipEventCreate(&ev[i]) != hipSuccess) return false;
    }
    ProfEntry pe;
    pe.name = name;
    pe.e0 = *e0 = ev[0]; pe.e1 = *e1 = ev[1];
    c->prof_pending.push_back(pe);
    return true;
}

void fd_prof_begin(const fdk::Launch &L, const char *name)
{
    fd_context *c = L.ctx;
    if (c->profile != 2 || L.capturing) return;
    ProfEntry pe;
    pe.name = name;
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        if (!c->event_pool.empty()) { ev[i] = c->event_pool.back(); c->event_pool.pop_back(); }
        else if (hipEventCreate(&ev[i]) != hipSuccess) return;
    }
    pe.e0 = ev[0]; pe.e1 = ev[1];
    hipEventRecord(pe.e0, L.stream);
    c->prof_pending.push_back(pe);
}

void fd_prof_end(const fdk::Launch &L)
{
    fd_context *c = L.ctx;
    if (c->profile != 2 || L.capturing || c->prof_pending.empty()) return;
    hipEventRecord(c->prof_pending.back().e1, L.stream);
}

void fd_prof_drain(fd_context *c)
{
    for (auto &pe : c->prof_pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(pe.e1) == hipSuccess && hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
            auto &acc = c->prof_acc[pe.name];
            acc.first += 1;
            acc.second += ms;
        }
        c->event_pool.push_back(pe.e0);
        c->event_pool.push_back(pe.e1);
    }
    c->prof_pending.clear();
}

// ------------------------------------------------------------------------------------------------
// expected state_dict (FastDiff_model.py:13-72; modules.py:116-125,141-187,257-318)
// ------------------------------------------------------------------------------------------------
struct ParamSpec { std::string name; std::vector<int64_t> dims; bool weight_norm; bool transposed_conv; bool linear; };

static const int KP_RES_IDX[6] = {1, 3, 6, 8, 11, 13};

// the state_dict of FastDiff(**cfg): names, shapes and registration facts (weight-normed Conv1d, plain ConvTranspose1d / Linear)
static std::vector<ParamSpec> param_specs(const fd_config &c)
{
    const int64_t C = c.inner_channels, COND = c.cond_channels, HID = c.kpnet_hidden_channels, KS = c.lvc_kernel_size, KK = c.kpnet_conv_size;
    const int64_t LAYERS = c.lvc_layers_each_block, E_IN = c.diffusion_step_embed_dim_in, E_MID = c.diffusion_step_embed_dim_mid, E_OUT = c.diffusion_step_embed_dim_out;
    std::vector<ParamSpec> s;
    s.push_back({"first_audio_conv", {C, 1, 7}, true, false, false});
    s.push_back({"fc_t1", {E_MID, E_IN}, false, false, true});
    s.push_back({"fc_t2", {E_OUT, E_MID}, false, false, true});
    for (int n = 0; n < c.n_upsample; ++n) {
        const std::string p = "lvc_blocks." + std::to_string(n);
        s.push_back({p + ".upsample", {C, C, 2 * (int64_t)c.upsample_ratios[n]}, false, true, false});
        s.push_back({p + ".kernel_predictor.input_conv.0", {HID, COND, 5}, true, false, false});
        for (int j = 0; j < 6; ++j)
            s.push_back({p + ".kernel_predictor.residual_conv." + std::to_string(KP_RES_IDX[j]), {HID, HID, KK}, true, false, false});
        s.push_back({p + ".kernel_predictor.kernel_conv", {LAYERS * C * 2 * C * KS, HID, KK}, true, false, false});
        s.push_back({p + ".kernel_predictor.bias_conv", {LAYERS * 2 * C, HID, KK}, true, false, false});
        s.push_back({p + ".fc_t", {COND, E_OUT}, false, false, true});
        for (int i = 0; i < LAYERS; ++i) s.push_back({p + ".convs." + std::to_string(i), {C, C, KS}, true, false, false});
        const std::string d = "downsample." + std::to_string(n);
        s.push_back({d + ".residual_dense", {C, C, 1}, true, false, false});
        for (int i = 0; i < 3; ++i) s.push_back({d + ".conv." + std::to_string(i), {C, C, 3}, true, false, false});
    }
    s.push_back({"final_conv.0", {c.audio_channels, C, 7}, true, false, false});
    return s;
}

static int64_t numel(const std::vector<int64_t> &d)
{
    int64_t n = 1;
    for (auto v : d) n *= v;
    return n;
}

// ------------------------------------------------------------------------------------------------
// C ABI: lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *fd_version(void) { return "fastdiff_hip 0.2 (gfx950)"; }
// 2: fd_sample settles its own range check before returning unless option defer_check = 1 (revision 1: always deferred)
int fd_abi_revision(void) { return 2; }

int fd_default_config(fd_config *cfg)
{
    if (!cfg) return FD_ERR_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->audio_channels = 1; cfg->inner_channels = 32; cfg->cond_channels = 80; cfg->n_upsample = 3;
    cfg->upsample_ratios[0] = 8; cfg->upsample_ratios[1] = 8; cfg->upsample_ratios[2] = 4;
    cfg->lvc_layers_each_block = 4; cfg->lvc_kernel_size = 3; cfg->kpnet_hidden_channels = 64; cfg->kpnet_conv_size = 3;
    cfg->diffusion_step_embed_dim_in = 128; cfg->diffusion_step_embed_dim_mid = 512; cfg->diffusion_step_embed_dim_out = 512;
    cfg->use_weight_norm = 1;
    return FD_OK;
}

const char *fd_last_error(fd_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void release_handle(fd_context *h);

int fd_create(const fd_config *cfg, int device, fd_handle *out)
{
    fd_context *nullh = nullptr;
    if (!cfg || !out) FD_FAIL(nullh, FD_ERR_INVALID, "fd_create: null argument");
    fd_config ref;
    fd_default_config(&ref);
    // base.yaml's architecture (every shipped YAML) runs on the tuned gfx950 kernel set; any other configuration the reference
    // constructor accepts (FastDiff_model.py:13-26) on the runtime-shaped kernels of fd_generic.hip
    bool is_base = cfg->audio_channels == ref.audio_channels && cfg->inner_channels == ref.inner_channels &&
                   cfg->cond_channels == ref.cond_channels && cfg->n_upsample == ref.n_upsample &&
                   cfg->lvc_layers_each_block == ref.lvc_layers_each_block && cfg->lvc_kernel_size == ref.lvc_kernel_size &&
                   cfg->kpnet_hidden_channels == ref.kpnet_hidden_channels && cfg->kpnet_conv_size == ref.kpnet_conv_size &&
                   cfg->diffusion_step_embed_dim_in == ref.diffusion_step_embed_dim_in &&
                   cfg->diffusion_step_embed_dim_mid == ref.diffusion_step_embed_dim_mid &&
                   cfg->diffusion_step_embed_dim_out == ref.diffusion_step_embed_dim_out;
    for (int i = 0; is_base && i < ref.n_upsample; ++i) is_base = cfg->upsample_ratios[i] == ref.upsample_ratios[i];
    if (!is_base) {
        std::string why;
        if (fdg::validate(*cfg, why) != FD_OK) FD_FAIL(nullh, FD_ERR_UNSUPPORTED, "fd_create: %s", why.c_str());
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        FD_FAIL(nullh, FD_ERR_HIP, "fd_create: no HIP device available (%s); there is no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= ndev) FD_FAIL(nullh, FD_ERR_INVALID, "fd_create: device %d out of range (0..%d)", device, ndev - 1);
    FD_HIP(nullh, hipSetDevice(device));
    fd_context *c = new fd_context();
    c->cfg = *cfg;
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
    for (int i = 0; i < ST_COUNT; ++i) c->fast[i] = true;
    // every resource of the handle lives in *c from the moment it exists, so one failure path (fd_destroy's own release code) frees
    // whatever was created before the failing call
    auto fail = [&](hipError_t err) {
        g_create_error = std::string("fd_create: ") + hipGetErrorString(err);
        release_handle(c);
        return FD_ERR_HIP;
    };
    if ((e = hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&c->scratch, 65536)) != hipSuccess) return fail(e);
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&c->flags_host), 256, hipHostMallocDefault)) != hipSuccess) return fail(e);
    memset(c->flags_host, 0, 256);
    if ((e = hipEventCreateWithFlags(&c->flags_done, hipEventDisableTiming)) != hipSuccess) return fail(e);
    if ((e = hipEventCreateWithFlags(&c->flags_done2, hipEventDisableTiming)) != hipSuccess) return fail(e);
    // the staging ring is allocated here (64 KB per slot covers the step table and a few thousand utterances): a call only
    // allocates pinned memory again for a larger batch than that
    for (auto &sl : c->stage) {
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&sl.host), 65536, hipHostMallocDefault)) != hipSuccess) return fail(e);
        sl.cap = 65536;
    }
    if (!is_base) fdg::create(c);
    *out = c;
    return FD_OK;
}

static void free_workspace(fd_context *c)
{
    Workspace &w = c->ws;
    void *ptrs[] = {w.noise, w.embed_h2, w.a[0], w.a[1], w.a[2], w.a[3], w.kp_h0, w.kp_hA, w.kp_hB, w.kpack, w.h_f16, w.range_flag, w.lens_dev, w.uid_dev, w.xsave, w.xA, w.xB,
                    w.xtap[0], w.xtap[1], w.xtap[2], w.mel, w.mel_rep, w.x, w.eps_acc, w.steps, w.params};
    for (void *p : ptrs)
        if (p) hipFree(p);
    w = Workspace();
}

// Destroys the retired graphs whose last launch has completed (all of them when `wait`: the caller has synchronised the device).
static void reap_retired(fd_context *c, bool wait)
{
    for (size_t i = 0; i < c->retired.size();) {
        fd_context::RetiredGraph &r = c->retired[i];
        if (!wait && r.done && hipEventQuery(r.done) != hipSuccess) { (void)hipGetLastError(); ++i; continue; }
        if (r.exec) hipGraphExecDestroy(r.exec);
        if (r.graph) hipGraphDestroy(r.graph);
        if (r.done) hipEventDestroy(r.done);
        c->retired.erase(c->retired.begin() + i);
    }
}

// Drops every captured graph.  synced: the caller has just synchronised the device (workspace growth, fd_commit_weights, fd_destroy);
// otherwise (an option that changes what a step launches) the device is synchronised here -- a replay may still be queued, and the
// stream it is queued on may be gone by now, so there is nothing to record an event on.
static void drop_graph(fd_context *c, bool synced = false)
{
    if (!synced && !(c->graphs.empty() && c->retired.empty())) hipDeviceSynchronize();
    for (auto &g : c->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
    reap_retired(c, true);
}

// Frees everything a handle owns (each member is null until created): the tail of fd_destroy and the failure path of fd_create.
static void release_handle(fd_context *h)
{
    fdg::destroy(h);
    if (h->flags_host) hipHostFree(h->flags_host);
    if (h->flags_done) hipEventDestroy(h->flags_done);
    if (h->flags_done2) hipEventDestroy(h->flags_done2);
    for (auto ev : h->event_pool) hipEventDestroy(ev);
    drop_graph(h, true);
    free_workspace(h);
    for (void *p : h->dev_allocs) hipFree(p);
    if (h->scratch) hipFree(h->scratch);
    if (h->lvc_scratch) hipFree(h->lvc_scratch);
    if (h->kconv_scratch) hipFree(h->kconv_scratch);
    if (h->cconv_scratch) hipFree(h->cconv_scratch);
    for (auto &sl : h->stage) {
        if (sl.host) hipHostFree(sl.host);
        if (sl.done) hipEventDestroy(sl.done);
    }
    for (void *p : h->mel_allocs) hipFree(p);
    if (h->ev_switch) hipEventDestroy(h->ev_switch);
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    delete h;
}

int fd_destroy(fd_handle h)
{
    if (!h) return FD_ERR_INVALID;
    hipSetDevice(h->device);
    fd_settle(h);
    hipDeviceSynchronize();
    fd_prof_drain(h);
    release_handle(h);
    return FD_OK;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
int fd_set_weight(fd_handle h, const char *name, const float *host_data, const int64_t *dims, int ndim)
{
    if (!h || !name || !host_data || !dims || ndim <= 0 || ndim > 4) return FD_ERR_INVALID;
    const std::string key(name);
    // find the owning parameter and the expected shape of this tensor
    const std::vector<ParamSpec> specs = param_specs(h->cfg);
    std::vector<int64_t> expect;
    for (const auto &s : specs) {
        if (key.compare(0, s.name.size(), s.name) != 0 || key.size() <= s.name.size() || key[s.name.size()] != '.') continue;
        const std::string suffix = key.substr(s.name.size() + 1);
        if (suffix == "weight" || suffix == "weight_v") expect = s.dims;
        else if (suffix == "weight_g") { expect = {s.dims[0], 1, 1}; }
        else if (suffix == "bias") expect = {s.transposed_conv ? s.dims[1] : s.dims[0]};
        else continue;
        break;
    }
    if (expect.empty()) FD_FAIL(h, FD_ERR_INVALID, "fd_set_weight: unexpected key '%s' (not in the FastDiff state_dict)", name);
    std::vector<int64_t> got(dims, dims + ndim);
    if (got != expect) {
        std::string a, b;
        for (auto v : got) a += std::to_string(v) + ",";
        for (auto v : expect) b += std::to_string(v) + ",";
        FD_FAIL(h, FD_ERR_INVALID, "fd_set_weight: size mismatch for %s: got [%s] expected [%s]", name, a.c_str(), b.c_str());
    }
    auto &slot = h->raw[key];
    slot.first = got;
    slot.second.assign(host_data, host_data + numel(got));
    h->committed = false;
    return FD_OK;
}

namespace {

typedef FoldedParam Folded;

// w = v * (g / ||v||), norm over everything but dim 0 (torch._weight_norm(v, g, 0)); plain weights pass through
int fold_param(fd_context *h, const ParamSpec &s, Folded &out)
{
    const auto itb = h->raw.find(s.name + ".bias");
    if (itb == h->raw.end()) FD_FAIL(h, FD_ERR_MISSING, "fd_commit_weights: missing tensor %s.bias", s.name.c_str());
    out.b = itb->second.second;
    const auto itw = h->raw.find(s.name + ".weight");
    const auto itv = h->raw.find(s.name + ".weight_v");
    const auto itg = h->raw.find(s.name + ".weight_g");
    if (itv != h->raw.end() && itg != h->raw.end()) {
        const std::vector<float> &v = itv->second.second, &g = itg->second.second;
        const int64_t cout = s.dims[0], per = numel(s.dims) / cout;
        out.w.resize(v.size());
        for (int64_t o = 0; o < cout; ++o) {
            double ss = 0.0;
            for (int64_t j = 0; j < per; ++j) ss += (double)v[o * per + j] * (double)v[o * per + j];
            const float scale = g[o] / (float)sqrt(ss);
            for (int64_t j = 0; j < per; ++j) out.w[o * per + j] = v[o * per + j] * scale;
        }
    } else if (itw != h->raw.end()) {
        out.w = itw->second.second;
    } else {
        FD_FAIL(h, FD_ERR_MISSING, "fd_commit_weights: missing tensor %s.weight (or weight_g/weight_v)", s.name.c_str());
    }
    return FD_OK;
}

int upload(fd_context *h, const void *src, size_t bytes, const void **dst)
{
    void *d = nullptr;
    FD_HIP(h, hipMalloc(&d, bytes));
    h->dev_allocs.push_back(d);
    FD_HIP(h, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return FD_OK;
}

int upload_f(fd_context *h, const std::vector<float> &v, const float **dst)
{
    return upload(h, v.data(), v.size() * sizeof(float), reinterpret_cast<const void **>(dst));
}

// Conv weight [cout][cin][ks] -> MFMA A-operand pack [mt][s4][lane][4], kk = tap*cin + ci = 2*(4*s4+r) + (lane>>5)
std::vector<float> pack_A(const std::vector<float> &w, int cout, int cin, int ks)
{
    const int ns4 = cin * ks / 8, nmt = cout / 32;
    std::vector<float> p((size_t)nmt * ns4 * 256);
    for (int mt = 0; mt < nmt; ++mt)
        for (int s4 = 0; s4 < ns4; ++s4)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int o = mt * 32 + (lane & 31), kk = 2 * (4 * s4 + r) + (lane >> 5);
                    const int tap = kk / cin, ci = kk % cin;
                    p[(((size_t)mt * ns4 + s4) * 64 + lane) * 4 + r] = w[((size_t)o * cin + ci) * ks + tap];
                }
    return p;
}

// IEEE binary16 <-> binary32 on the host (round to nearest even, subnormals kept): the weight pieces of the fp16x2 GEMM.
static uint16_t f16_from_f32(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7FFFFFFFu;
    if (u > 0x7F800000u) return sign | 0x7E00u;                  // NaN
    if (u >= 0x477FF000u) return sign | 0x7C00u;                 // >= 65520 rounds to infinity
    if (u < 0x38800000u) {                                       // below 2^-14: subnormal, a multiple of 2^-24
        float ax;
        memcpy(&ax, &u, 4);
        return sign | (uint16_t)lrintf(ax * 16777216.0f);        // current rounding mode = nearest even; 1024 = smallest normal
    }
    uint32_t hbits = (((u >> 23) - 112u) << 10) | ((u & 0x7FFFFFu) >> 13);
    const uint32_t rem = u & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (hbits & 1u))) ++hbits;   // a carry into the exponent is the correct result
    return sign | (uint16_t)hbits;
}
static float f32_from_f16(uint16_t hb)
{
    const uint32_t sign = (uint32_t)(hb & 0x8000u) << 16, exp = (hb >> 10) & 0x1Fu, man = hb & 0x3FFu;
    float v;
    if (exp == 0) v = (float)man * (1.0f / 16777216.0f);
    else if (exp == 31) { const uint32_t u = 0x7F800000u | (man << 13); memcpy(&v, &u, 4); }
    else { const uint32_t u = ((exp + 112u) << 23) | (man << 13); memcpy(&v, &u, 4); }
    uint32_t u;
    memcpy(&u, &v, 4);
    u |= sign;
    memcpy(&v, &u, 4);
    return v;
}

// fp16 pieces of a conv weight [cout][cin][ks] (cout a multiple of 32) in 32x32x16 A-operand order:
// [mt = out/32][piece][kg][lane = out%32 + 32*g][8], k = 16*kg + 8*g + e = tap*cin + in.  *ok is cleared when a value does not
// fit the fp16 range.
static std::vector<uint16_t> pack_A_h2(const std::vector<float> &w, int cin, int ks, bool *ok, int cout = 32)
{
    const int nk = cin * ks, nkg = nk / 16, nmt = cout / 32;
    std::vector<uint16_t> hp((size_t)nmt * 2 * nkg * 64 * 8);
    for (int mt = 0; mt < nmt; ++mt)
        for (int kg = 0; kg < nkg; ++kg)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int kk = kg * 16 + 8 * (lane >> 5) + e, tap = kk / cin, in = kk % cin, out = mt * 32 + (lane & 31);
                    const float v = w[((size_t)out * cin + in) * ks + tap];
                    if (!(fabsf(v) < 32768.0f)) *ok = false;
                    const uint16_t p1 = f16_from_f32(v);
                    hp[((((size_t)mt * 2 + 0) * nkg + kg) * 64 + lane) * 8 + e] = p1;
                    hp[((((size_t)mt * 2 + 1) * nkg + kg) * 64 + lane) * 8 + e] = f16_from_f32((v - f32_from_f16(p1)) * 2048.0f);
                }
    return hp;
}

void unpack_kernel_index(int p, int &layer, int &in, int &out, int &tap)
{
    layer = p / fd::KLAYER;
    const int q = p % fd::KLAYER, e = q & 7, lane = (q >> 3) & 63, mk = q >> 9;
    const int mt = mk / 6, kg = mk % 6, kk = kg * 16 + 8 * (lane >> 5) + e, row = lane & 31;
    tap = kk / fd::C; in = kk % fd::C;
    out = 16 * mt + (row & 15) + 32 * (row >> 4);      // inverse of kernel_tile / kernel_row
}

}  // namespace

int fd_commit_weights(fd_handle h)
{
    if (!h) return FD_ERR_INVALID;
    FD_HIP(h, hipSetDevice(h->device));
    {
        const int rcs = fd_settle(h);      // a pending host check would otherwise run its call again on the NEW weights
        if (rcs != FD_OK) return rcs;
    }
    h->embed_valid = false;
    FD_HIP(h, hipDeviceSynchronize());
    h->committed = false;            // until the new set is complete: a failed re-commit must not leave the old flag over freed weights
    for (void *p : h->dev_allocs) hipFree(p);
    h->dev_allocs.clear();
    h->w = DevWeights();
    drop_graph(h, true);

    std::map<std::string, Folded> f;
    for (const auto &s : param_specs(h->cfg)) {
        int rc = fold_param(h, s, f[s.name]);
        if (rc != FD_OK) return rc;
    }
    if (h->gen) {      // another architecture than base.yaml's: folded reference-layout weights, no operand packing
        const int rcg = fdg::commit(h, f);
        if (rcg == FD_OK) h->committed = true;
        return rcg;
    }
    DevWeights &w = h->w;
    int rc;
#define UP(vec, dst) if ((rc = upload_f(h, vec, &(dst))) != FD_OK) return rc
    auto up_conv = [&](const std::string &name, ConvW &cw) -> int {
        int r1 = upload_f(h, f[name].w, &cw.w);
        if (r1 != FD_OK) return r1;
        return upload_f(h, f[name].b, &cw.b);
    };
    if ((rc = up_conv("first_audio_conv", w.first)) != FD_OK) return rc;
    if ((rc = up_conv("final_conv.0", w.final_)) != FD_OK) return rc;
    {   // the same weights in the order the last LVC layer holds its outputs: channel = 16 mt + 4 hi + (r & 3) + 8 (r >> 2)
        const std::vector<float> &fw = f["final_conv.0"].w;
        std::vector<float> ff(4 * 8 * 8, 0.0f);
        for (int part = 0; part < 4; ++part)
            for (int r = 0; r < 8; ++r)
                for (int k = 0; k < 7; ++k) ff[(part * 8 + r) * 8 + k] = fw[(16 * (part >> 1) + 4 * (part & 1) + (r & 3) + 8 * (r >> 2)) * 7 + k];
        UP(ff, w.final_fuse);
    }
    // embed MLP, transposed
    auto transpose = [](const std::vector<float> &m, int rows, int cols) {
        std::vector<float> t((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = m[(size_t)r * cols + c];
        return t;
    };
    UP(transpose(f["fc_t1"].w, fd::E_MID, fd::E_IN), w.fc_t1_T);
    UP(f["fc_t1"].b, w.fc_t1_b);
    UP(transpose(f["fc_t2"].w, fd::E_OUT, fd::E_MID), w.fc_t2_T);
    UP(f["fc_t2"].b, w.fc_t2_b);
    {   // frequency table of calc_diffusion_step_embedding (util.py:425-427): fp32 product, fp32 exp
        std::vector<float> table(64);
        const float cst = (float)(-(log(10000.0) / 63.0));
        for (int j = 0; j < 64; ++j) {
            volatile float arg = (float)j * cst;
            table[j] = expf(arg);
        }
        UP(table, w.embed_table);
    }
    bool f16_ok = true, w_ok = true, lvc_ok = true, dblock_ok = true, convt_ok = true, kpf_ok = true;
    for (int n = 0; n < fd::NBLK; ++n) {
        const std::string p = "lvc_blocks." + std::to_string(n), d = "downsample." + std::to_string(n);
        if ((rc = up_conv(d + ".residual_dense", w.down[n].res)) != FD_OK) return rc;
        for (int i = 0; i < 3; ++i) {
            if ((rc = up_conv(d + ".conv." + std::to_string(i), w.down[n].conv[i])) != FD_OK) return rc;
            UP(pack_A(f[d + ".conv." + std::to_string(i)].w, fd::C, fd::C, 3), w.down_pack[n][i]);
        }
        UP(pack_A(f[d + ".residual_dense"].w, fd::C, fd::C, 1), w.down_pack[n][3]);
        for (int i = 0; i < 4; ++i) {      // the same four matrices as fp16 pieces (conv 0..2: K = 96, residual 1x1: K = 32)
            const std::vector<uint16_t> hp = pack_A_h2(f[i < 3 ? d + ".conv." + std::to_string(i) : d + ".residual_dense"].w, fd::C, i < 3 ? 3 : 1, &dblock_ok);
            if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.down_h2[n][i]))) != FD_OK) return rc;
        }
        if ((rc = up_conv(p + ".fc_t", w.blk[n].fc_t)) != FD_OK) return rc;
        UP(transpose(f[p + ".fc_t"].w, fd::COND, fd::E_OUT), w.fc_t_T[n]);
        w.fc_t_b[n] = w.blk[n].fc_t.b;
        if ((rc = up_conv(p + ".upsample", w.blk[n].up)) != FD_OK) return rc;
        {   // ConvTranspose1d weight [in][out][2r] -> per-phase MFMA A operands [ph][s4][lane][4], kk = sel*32 + i
            const int r = fd::ratio(n), ks = 2 * r, pd = r / 2;
            const std::vector<float> &uw = f[p + ".upsample"].w;
            std::vector<float> up((size_t)r * 8 * 256);
            for (int ph = 0; ph < r; ++ph)
                for (int s4 = 0; s4 < 8; ++s4)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q) {
                            const int kk = 2 * (4 * s4 + q) + (lane >> 5), sel = kk >> 5, i = kk & 31, o = lane & 31;
                            // sel 0: the nearer input position (jA), sel 1: the one before it (jB = jA - 1, tap + r)
                            const int kA = (ph < pd) ? ph + pd : ph - pd, k = sel ? kA + r : kA;
                            up[(((size_t)ph * 8 + s4) * 64 + lane) * 4 + q] = uw[((size_t)i * fd::C + o) * ks + k];
                        }
            UP(up, w.up_pack[n]);
            // the same per-phase slices as fp16 pieces: [ph][piece][4 kg][64 lane = out + 32*g][8], k = 16*kg + 8*g + e = sel*32 + i
            std::vector<uint16_t> hp((size_t)r * 2 * 4 * 64 * 8);
            for (int ph = 0; ph < r; ++ph)
                for (int kg = 0; kg < 4; ++kg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int kk = kg * 16 + 8 * (lane >> 5) + e, sel = kk >> 5, i = kk & 31, o = lane & 31;
                            const int kA = (ph < pd) ? ph + pd : ph - pd, k = sel ? kA + r : kA;
                            const float v = uw[((size_t)i * fd::C + o) * ks + k];
                            if (!(fabsf(v) < 32768.0f)) convt_ok = false;
                            const uint16_t p1 = f16_from_f32(v);
                            hp[((((size_t)ph * 2 + 0) * 4 + kg) * 64 + lane) * 8 + e] = p1;
                            hp[((((size_t)ph * 2 + 1) * 4 + kg) * 64 + lane) * 8 + e] = f16_from_f32((v - f32_from_f16(p1)) * 2048.0f);
                        }
            if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.up_h2[n]))) != FD_OK) return rc;
        }
        if ((rc = up_conv(p + ".kernel_predictor.input_conv.0", w.blk[n].kp_in)) != FD_OK) return rc;
        UP(pack_A(f[p + ".kernel_predictor.input_conv.0"].w, fd::HID, fd::COND, 5), w.kp_in_pack[n]);
        {
            const std::vector<uint16_t> hp = pack_A_h2(f[p + ".kernel_predictor.input_conv.0"].w, fd::COND, 5, &kpf_ok, fd::HID);
            if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.kp_in_h2[n]))) != FD_OK) return rc;
        }
        for (int j = 0; j < 6; ++j) {
            const std::string nm = p + ".kernel_predictor.residual_conv." + std::to_string(KP_RES_IDX[j]);
            if ((rc = up_conv(nm, w.blk[n].kp_res[j])) != FD_OK) return rc;
            UP(pack_A(f[nm].w, fd::HID, fd::HID, 3), w.kp_res_pack[n][j]);
            const std::vector<uint16_t> hp = pack_A_h2(f[nm].w, fd::HID, 3, &kpf_ok, fd::HID);
            if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.kp_res_h2[n][j]))) != FD_OK) return rc;
        }
        if ((rc = up_conv(p + ".kernel_predictor.kernel_conv", w.blk[n].kc)) != FD_OK) return rc;
        if ((rc = up_conv(p + ".kernel_predictor.bias_conv", w.blk[n].bc)) != FD_OK) return rc;
        for (int i = 0; i < fd::LAYERS; ++i) {
            if ((rc = up_conv(p + ".convs." + std::to_string(i), w.blk[n].convs[i])) != FD_OK) return rc;
            UP(pack_A(f[p + ".convs." + std::to_string(i)].w, fd::C, fd::C, 3), w.lvc_conv_pack[n][i]);
            if (n == 0) {      // hop 8: 16x16x32 tiles: lane = out%16 + 16*g holds k = tap*32 + 8g + e, i.e. input channels 8g .. 8g+7 of one tap
                const std::vector<float> &cw = f[p + ".convs." + std::to_string(i)].w;
                std::vector<uint16_t> hp((size_t)2 * 3 * 2 * 64 * 8);
                for (int rt = 0; rt < 2; ++rt)
                    for (int tap = 0; tap < 3; ++tap)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int out = 16 * rt + (lane & 15), in = 8 * (lane >> 4) + e;
                                const float v = cw[((size_t)out * fd::C + in) * 3 + tap];
                                if (!(fabsf(v) < 32768.0f)) lvc_ok = false;
                                const uint16_t p1 = f16_from_f32(v);
                                hp[((((size_t)rt * 3 + tap) * 2 + 0) * 64 + lane) * 8 + e] = p1;
                                hp[((((size_t)rt * 3 + tap) * 2 + 1) * 64 + lane) * 8 + e] = f16_from_f32((v - f32_from_f16(p1)) * 2048.0f);
                            }
                if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.lvc_conv_h16[i]))) != FD_OK) return rc;
            }
            {
                const std::vector<uint16_t> hp = pack_A_h2(f[p + ".convs." + std::to_string(i)].w, fd::C, 3, &lvc_ok);
                if ((rc = upload(h, hp.data(), hp.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.lvc_conv_h2[n][i]))) != FD_OK)
                    return rc;
            }
        }
        {   // GEMM B-operand pack: rows in packed-record order, kk = tap*64 + c
            const std::vector<float> &kc = f[p + ".kernel_predictor.kernel_conv"].w, &kcb = f[p + ".kernel_predictor.kernel_conv"].b;
            const std::vector<float> &bc = f[p + ".kernel_predictor.bias_conv"].w, &bcb = f[p + ".kernel_predictor.bias_conv"].b;
            std::vector<float> gp((size_t)(fd::KREC / 32) * 24 * 256), gb(fd::KREC);
            for (int pt = 0; pt < fd::KREC / 32; ++pt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int pp = pt * 32 + (lane & 31);
                    const float *wrow;
                    if (pp < fd::KW) {
                        int layer, in, out, tap;
                        unpack_kernel_index(pp, layer, in, out, tap);
                        const int row = ((layer * fd::C + in) * 2 * fd::C + out) * 3 + tap;   // [layers,in,out,k] view (modules.py:333-338)
                        wrow = kc.data() + (size_t)row * fd::HID * 3;
                        gb[pp] = kcb[row];
                    } else {
                        // bias record [layer][mt][row]  ->  bias_conv row layer*64 + out (view [layers,out], modules.py:339-342)
                        const int q = pp - fd::KW, layer = q >> 6, mt = (q >> 5) & 1, row = q & 31;
                        const int brow = layer * 64 + 16 * mt + (row & 15) + 32 * (row >> 4);
                        wrow = bc.data() + (size_t)brow * fd::HID * 3;
                        gb[pp] = bcb[brow];
                    }
                    for (int s4 = 0; s4 < 24; ++s4)
                        for (int r = 0; r < 4; ++r) {
                            const int kk = 2 * (4 * s4 + r) + (lane >> 5), tap = kk / fd::HID, c = kk % fd::HID;
                            gp[(((size_t)pt * 24 + s4) * 64 + lane) * 4 + r] = wrow[c * 3 + tap];
                        }
                }
            UP(gp, w.gemm_pack[n]);
            UP(gb, w.gemm_bias[n]);
            // fp16x2 form: w = w1 + 2^-11 * w2, w1 = fp16(w), w2 = fp16((w - w1) * 2^11) (round to nearest even, subnormals kept);
            // B operand of v_mfma_f32_32x32x16_f16: lane = col + 32*g holds the 8 consecutive k = kg*16 + 8g + e, k = tap*64 + channel
            std::vector<uint16_t> gx((size_t)(fd::KREC / 32) * 2 * 12 * 64 * 8);
            for (int pt = 0; pt < fd::KREC / 32; ++pt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int pp = pt * 32 + (lane & 31), g = lane >> 5;
                    const float *wrow;
                    if (pp < fd::KW) {
                        int layer, in, out, tap;
                        unpack_kernel_index(pp, layer, in, out, tap);
                        wrow = kc.data() + (size_t)(((layer * fd::C + in) * 2 * fd::C + out) * 3 + tap) * fd::HID * 3;
                    } else {
                        const int q = pp - fd::KW, layer = q >> 6, mt = (q >> 5) & 1, row = q & 31;
                        wrow = bc.data() + (size_t)(layer * 64 + 16 * mt + (row & 15) + 32 * (row >> 4)) * fd::HID * 3;
                    }
                    for (int kg = 0; kg < 12; ++kg)
                        for (int e = 0; e < 8; ++e) {
                            const int kk = kg * 16 + g * 8 + e, tap = kk / fd::HID, ch = kk % fd::HID;
                            const float v = wrow[ch * 3 + tap];
                            if (!(fabsf(v) < 32768.0f)) f16_ok = false;
                            const uint16_t p1 = f16_from_f32(v);
                            const uint16_t p2 = f16_from_f32((v - f32_from_f16(p1)) * 2048.0f);
                            gx[((((size_t)pt * 2 + 0) * 12 + kg) * 64 + lane) * 8 + e] = p1;
                            gx[((((size_t)pt * 2 + 1) * 12 + kg) * 64 + lane) * 8 + e] = p2;
                        }
                }
            if ((rc = upload(h, gx.data(), gx.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.gemm_h2_pack[n]))) != FD_OK)
                return rc;
            // Winograd F(2,3) over the frame axis (kernel_conv is a k = 3 convolution over frames, modules.py:315-318): per pair of
            // output frames  y[2p] = m0 + m1 + m2,  y[2p+1] = m1 - m2 + m3  with  m_j = V_j . u_j (K = 64 each),
            //   V0 = g0, V1 = (g0 + g1 + g2) / 2, V2 = (g0 - g1 + g2) / 2, V3 = -g2        (g_tap = the column's weights of that tap)
            //   u0 = h[2p-1] - h[2p+1], u1 = h[2p] + h[2p+1], u2 = h[2p+1] - h[2p], u3 = h[2p] - h[2p+2]   (k_h_wino)
            // B operand: lane = col + 32*g holds the 8 consecutive k = kg*16 + 8g + e, kg = 4 j + k4, channel = 16 k4 + 8 g + e
            std::vector<uint16_t> gw((size_t)(fd::KREC / 32) * 2 * 16 * 64 * 8);
            for (int pt = 0; pt < fd::KREC / 32; ++pt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int pp = pt * 32 + (lane & 31), g = lane >> 5;
                    const float *wrow;
                    if (pp < fd::KW) {
                        int layer, in, out, tap;
                        unpack_kernel_index(pp, layer, in, out, tap);
                        wrow = kc.data() + (size_t)(((layer * fd::C + in) * 2 * fd::C + out) * 3 + tap) * fd::HID * 3;
                    } else {
                        const int q = pp - fd::KW, layer = q >> 6, mt = (q >> 5) & 1, row = q & 31;
                        wrow = bc.data() + (size_t)(layer * 64 + 16 * mt + (row & 15) + 32 * (row >> 4)) * fd::HID * 3;
                    }
                    for (int kg = 0; kg < 16; ++kg)
                        for (int e = 0; e < 8; ++e) {
                            const int j = kg >> 2, ch = (kg & 3) * 16 + g * 8 + e;
                            const double g0 = wrow[ch * 3 + 0], g1 = wrow[ch * 3 + 1], g2 = wrow[ch * 3 + 2];
                            const float v = (float)(j == 0 ? g0 : (j == 1 ? 0.5 * (g0 + g1 + g2) : (j == 2 ? 0.5 * (g0 - g1 + g2) : -g2)));
                            if (!(fabsf(v) < 32768.0f)) w_ok = false;
                            const uint16_t p1 = f16_from_f32(v);
                            const uint16_t p2 = f16_from_f32((v - f32_from_f16(p1)) * 2048.0f);
                            gw[((((size_t)pt * 2 + 0) * 16 + kg) * 64 + lane) * 8 + e] = p1;
                            gw[((((size_t)pt * 2 + 1) * 16 + kg) * 64 + lane) * 8 + e] = p2;
                        }
                }
            if ((rc = upload(h, gw.data(), gw.size() * sizeof(uint16_t), reinterpret_cast<const void **>(&w.gemm_w_pack[n]))) != FD_OK)
                return rc;
        }
    }
    w.gemm_f16_ok = f16_ok;
    w.gemm_w_ok = w_ok;
    w.lvc_f16_ok = lvc_ok;
    w.dblock_f16_ok = dblock_ok;
    w.convt_f16_ok = convt_ok;
    w.kpf_f16_ok = kpf_ok;
    {
        std::vector<int> perm(fd::KW);
        for (int layer = 0; layer < fd::LAYERS; ++layer)
            for (int in = 0; in < fd::C; ++in)
                for (int out = 0; out < 2 * fd::C; ++out)
                    for (int tap = 0; tap < 3; ++tap)
                        perm[((layer * fd::C + in) * 2 * fd::C + out) * 3 + tap] = fd::kernel_index(layer, in, out, tap);
        if ((rc = upload(h, perm.data(), perm.size() * sizeof(int), reinterpret_cast<const void **>(&w.kc_perm))) != FD_OK) return rc;
        std::vector<int> bperm(fd::KB);
        for (int layer = 0; layer < fd::LAYERS; ++layer)
            for (int out = 0; out < 2 * fd::C; ++out) bperm[layer * 64 + out] = fd::bias_index(layer, out) - fd::KW;
        if ((rc = upload(h, bperm.data(), bperm.size() * sizeof(int), reinterpret_cast<const void **>(&w.bc_perm))) != FD_OK) return rc;
    }
#undef UP
    h->raw.clear();     // host copies are no longer needed
    h->committed = true;
    return FD_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
// Buffers scale with three quantities of a call: B (per-utterance arrays), B*T (every activation, the predicted kernels) and
// B*gx_rows(T) (the GEMM's fp16 image, padded per utterance).  Capacity is tracked in exactly those terms, so a handle that served
// (B=64, T=864) and then meets (B=1, T=20000) needs room for max(64*864, 20000) frames -- not for 64 x 20000.
static hipError_t allocate_workspace(fd_context *h, int64_t capB, int64_t frames, int64_t rows, int64_t pframes, int64_t prows, int64_t plens,
                                     size_t *total_out)
{
    h->embed_valid = false;          // a fresh noise table
    Workspace &w = h->ws;
    const size_t f = sizeof(float), FL = (size_t)frames * fd::HOPT;      // FL: samples of all utterances together
    size_t total = 0;
    auto alloc = [&](float **p, size_t n) -> hipError_t {
        total += n * f;
        hipError_t e1 = hipMalloc(reinterpret_cast<void **>(p), n * f);
        // zero once: with ragged batches (lens) tiles behind an utterance are never written, and nothing a later kernel
        // stages next to them (range checks run over whole tiles) should meet NaN bit patterns of a fresh allocation
        return e1 == hipSuccess ? hipMemset(*p, 0, n * f) : e1;
    };
    hipError_t e = hipSuccess;
#define WS(p, n) if (e == hipSuccess) e = alloc(&(p), (n))
    WS(w.noise, (size_t)1024 * capB * fd::NBLK * fd::COND);
    WS(w.embed_h2, (size_t)std::max<int64_t>(1024, capB) * fd::E_OUT);
    WS(w.a[0], fd::C * FL); WS(w.a[1], fd::C * FL / 4); WS(w.a[2], fd::C * FL / 32); WS(w.a[3], (size_t)fd::C * frames);
    // the predictor's buffers: their own capacities (a hoisted predictor holds N reverse steps: fd_internal.h)
    WS(w.kp_h0, (size_t)fd::NBLK * fd::HID * pframes); WS(w.kp_hA, (size_t)fd::NBLK * fd::HID * pframes);
    WS(w.kp_hB, (size_t)fd::NBLK * fd::HID * pframes);
    WS(w.kpack, (size_t)fd::NBLK * pframes * fd::KREC);
    WS(w.h_f16, (size_t)fd::NBLK * prows * 64 + 1024);      // + slack for the rounded-up last DMA
    WS(w.mel_rep, (size_t)fd::COND * pframes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&w.lens_dev), sizeof(int) * (size_t)std::max<int64_t>(plens, 64));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&w.uid_dev), sizeof(unsigned long long) * (size_t)std::max<int64_t>(capB, 64));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&w.range_flag), 512);
    if (e == hipSuccess) e = hipMemset(w.range_flag, 0, 512);
    WS(w.xA, fd::C * FL); WS(w.xB, fd::C * FL);
    WS(w.xtap[0], fd::C * FL / 32); WS(w.xtap[1], fd::C * FL / 4); WS(w.xtap[2], fd::C * FL);
    WS(w.mel, (size_t)fd::COND * frames); WS(w.x, FL); WS(w.xsave, FL); WS(w.eps_acc, FL); WS(w.steps, (size_t)std::max<int64_t>(capB, 64));
#undef WS
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&w.params), sizeof(StepParams));
    if (e == hipSuccess) e = hipMemset(w.params, 0, sizeof(StepParams));
    *total_out = total;
    return e;
}

static bool workspace_fits(const fd_context *h, int B, int T, int pmult)
{
    const Workspace &w = h->ws;
    const int64_t frames = (int64_t)B * T, rows = (int64_t)B * gx_rows_host(T);
    return w.B >= B && w.frames >= frames && w.rows >= rows && w.params && w.pframes >= frames * pmult && w.prows >= rows * pmult &&
           w.plens >= (int64_t)B * pmult;
}

static int ensure_workspace(fd_context *h, int B, int T, int pmult = 1)
{
    Workspace &w = h->ws;
    const int64_t frames = (int64_t)B * T, rows = (int64_t)B * gx_rows_host(T);
    if (workspace_fits(h, B, T, pmult)) return FD_OK;
    FD_HIP(h, hipDeviceSynchronize());
    drop_graph(h, true);
    // grow to the largest of each quantity seen so far, so that alternating shapes settle; if that does not fit, this call's own
    // needs alone are tried before giving up
    const int64_t own[6] = {B, frames, rows, frames * pmult, rows * pmult, (int64_t)B * pmult};
    const int64_t seen[6] = {w.B, w.frames, w.rows, w.pframes, w.prows, w.plens};
    int64_t want[2][6];
    bool same = true;
    for (int i = 0; i < 6; ++i) {
        want[0][i] = std::max(own[i], seen[i]);
        // a stream of requests of growing length (the reference CLI's pattern) would otherwise re-allocate -- a device synchronisation,
        // every buffer freed and allocated again, every graph dropped -- once per new maximum: grow by at least a quarter
        if (own[i] > seen[i] && seen[i] > 0) want[0][i] = std::max(own[i], seen[i] + seen[i] / 4);
        want[1][i] = own[i];
        same = same && want[0][i] == want[1][i];
    }
    size_t total = 0;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 2; ++attempt) {
        free_workspace(h);
        e = allocate_workspace(h, want[attempt][0], want[attempt][1], want[attempt][2], want[attempt][3], want[attempt][4], want[attempt][5], &total);
        if (e == hipSuccess) {
            w.B = (int)want[attempt][0]; w.frames = want[attempt][1]; w.rows = want[attempt][2];
            w.pframes = want[attempt][3]; w.prows = want[attempt][4]; w.plens = want[attempt][5]; w.bytes = total;
            return FD_OK;
        }
        (void)hipGetLastError();
        if (same) break;
    }
    free_workspace(h);
    FD_FAIL(h, FD_ERR_HIP, "workspace allocation for B=%d T=%d (%.1f MB) failed: %s", B, T, total / 1e6, hipGetErrorString(e));
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// stage dispatch + the denoiser step
// ------------------------------------------------------------------------------------------------
namespace fdk {

hipError_t first_conv(const Launch &L, const StepIO &io, int B, int T)
{
    return L.ctx->fast[ST_FIRST] ? fast_first_conv(L, io, B, T) : naive_first_conv(L, io, B, T);
}
hipError_t dblock(const Launch &L, const StepIO &io, int d, int B, int T)
{
    return L.ctx->fast[ST_DBLOCK] ? fast_dblock(L, d, B, T, io.x_in) : naive_dblock(L, d, B, T);
}
hipError_t kp_front(const Launch &L, const StepIO &io, int B, int T)
{
    L.ctx->h_image_ready = false;      // only the fp16-pipe front writes the GEMM's h image itself
    return L.ctx->fast[ST_KP_FRONT] ? fast_kp_front(L, io, B, T) : naive_kp_front(L, io, B, T);
}
hipError_t kp_gemm(const Launch &L, int B, int T) { return L.ctx->fast[ST_KP_GEMM] ? fast_kp_gemm(L, B, T) : naive_kp_gemm(L, B, T); }

// One TimeAware_LVCBlock (modules.py:190-218) given the packed kernels of kp_gemm.  x_in: [B,32,Lin].
static hipError_t lvc_block_run(const Launch &L, int n, const float *x_in, int B, int T, float **x_out)
{
    fd_context *c = L.ctx;
    Workspace &ws = c->ws;
    const int Lin = T * (fd::hop(n) / fd::ratio(n));
    float *cur = (x_in == ws.xA) ? ws.xB : ws.xA;
    float *other = (cur == ws.xA) ? ws.xB : ws.xA;
    const float *skip = ws.a[2 - n];
    // blocks 1 and 2 (hop 64, 256) on the fp16x2 pipe with the range check on the host: the up-sampler runs inside the first layer
    // (k_lvc_h2<.., UP>), its output never goes to HBM and back; x_in is read by that layer, which writes the other buffer
    const bool fuse_up = c->fuse_up && n >= 1 && c->fast[ST_CONVT] && c->fast[ST_LVC] &&
                         fd_pipe(c, c->conv_f16 && c->w.convt_f16_ok, 16 + n) == PIPE_F16_ONLY &&
                         fd_pipe(c, c->lvc_f16 && c->w.lvc_f16_ok, 1 + n * fd::LAYERS) == PIPE_F16_ONLY;
    hipError_t e = hipSuccess;
    if (fuse_up) {
        if ((e = fast_lvc_layer(L, n, 0, x_in, skip, cur, B, T, true)) != hipSuccess) return e;
    } else {
        e = c->fast[ST_CONVT] ? fast_convt(L, n, x_in, cur, B, Lin) : naive_convt(L, n, x_in, cur, B, Lin);
        if (e != hipSuccess) return e;
    }
    for (int i = fuse_up ? 1 : 0; i < fd::LAYERS; ++i) {
        if (c->fast[ST_LVC]) {
            e = fast_lvc_layer(L, n, i, cur, skip, other, B, T);
            std::swap(cur, other);
        } else {
            e = naive_lvc_layer(L, n, i, cur, skip, other, B, T);
        }
        if (e != hipSuccess) return e;
    }
    if (c->keep_taps) {
        e = hipMemcpyAsync(ws.xtap[n], cur, sizeof(float) * (size_t)B * fd::C * T * fd::hop(n), hipMemcpyDeviceToDevice, L.stream);
        if (e != hipSuccess) return e;
    }
    *x_out = cur;
    return hipSuccess;
}

static hipError_t run_step(const Launch &L, const StepIO &io, int B, int T)
{
    fd_context *c = L.ctx;
    Workspace &ws = c->ws;
    hipError_t e;
    // the reference's order of statements: down path (first conv, DBlocks), predictor (front + GEMM), the three LVC blocks.  With a
    // hoisted predictor (hoist_np > 1) front + GEMM of all N steps ran in front of the loop (sample_core).  Other orders and a second
    // stream were measured and did not pay (LABBOOK.md: overlap = gemm | paths, order = split | predictor).
    const bool hoisted = c->hoist_np > 1;
    // (round 6, measured and not kept: first_conv -- whose output a0 has no reader before the last block -- on a side branch of the
    // graph next to the DBlocks, joined in front of block 2, bit-identical: B=8 +0.5 %, B=1 +6 %; LABBOOK R6.7)
    if ((e = first_conv(L, io, B, T)) != hipSuccess) return e;
    for (int d = 0; d < fd::NBLK; ++d)
        if ((e = dblock(L, io, d, B, T)) != hipSuccess) return e;
    if (!hoisted) {
        if ((e = kp_front(L, io, B, T)) != hipSuccess) return e;
        if ((e = kp_gemm(L, B, T)) != hipSuccess) return e;
    }
    float *x = ws.a[3];
    for (int n = 0; n < fd::NBLK; ++n) {
        float *xo = nullptr;
        if ((e = lvc_block_run(L, n, x, B, T, &xo)) != hipSuccess) return e;
        x = xo;
    }
    if (c->fast[ST_FINAL]) return fast_final(L, io, x, B, T);
    // naive tail: eps into the free ping-pong buffer (or the caller's), then the separate update kernel
    float *eps = io.sampler ? ((x == ws.xA) ? ws.xB : ws.xA) : io.eps_out;
    if ((e = naive_final_eps(L, x, eps, B, T)) != hipSuccess) return e;
    if (io.sampler) return naive_update(L, ws.x, eps, (int64_t)B * T * fd::HOPT);
    return hipSuccess;
}

}  // namespace fdk

// ------------------------------------------------------------------------------------------------
// C ABI: compute
// ------------------------------------------------------------------------------------------------
extern "C" {


static int check_common(fd_handle h, int B, int T, const char *who)
{
    if (!h) return FD_ERR_INVALID;
    if (!h->committed) FD_FAIL(h, FD_ERR_STATE, "%s: weights not committed (call fd_commit_weights after fd_set_weight)", who);
    if (B <= 0 || T <= 0) FD_FAIL(h, FD_ERR_INVALID, "%s: B=%d T=%d must be positive", who, B, T);
    if ((int64_t)B * T * fdg::hop_total(h) * (h->gen ? h->cfg.inner_channels : fd::C) >= (int64_t)1 << 31)
        FD_FAIL(h, FD_ERR_INVALID, "%s: B*T too large for one call (B=%d, T=%d); split the batch", who, B, T);
    FD_HIP(h, hipSetDevice(h->device));
    return FD_OK;
}

// Pinned staging (fd_context::stage): the next slot of the ring with room for `bytes`, free to be written by the host.
int fd_stage_acquire(fd_handle h, size_t bytes, fd_context::StageSlot **out)
{
    fd_context::StageSlot &sl = h->stage[h->stage_next++ % fd_context::STAGE_SLOTS];
    if (sl.done) FD_HIP(h, hipEventSynchronize(sl.done));             // the uploads that last used this slot (8 calls ago)
    else FD_HIP(h, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (sl.cap < bytes) {
        if (sl.host) hipHostFree(sl.host);
        sl.host = nullptr; sl.cap = 0;
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        FD_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&sl.host), cap, hipHostMallocDefault));
        sl.cap = cap;
    }
    *out = &sl;
    return FD_OK;
}
// ... and the mark behind the copies that read it
int fd_stage_commit(fd_handle h, fd_context::StageSlot *sl, hipStream_t stream)
{
    FD_HIP(h, hipEventRecord(sl->done, stream));
    return FD_OK;
}

// One stream at a time per handle: a call on another stream than the previous one first settles what is pending there and then makes
// the new stream wait for the tail of the handle's last call (workspace, embedding rows and step parameters are reused from call to
// call).  The tail is an EVENT recorded at the end of every call (mark_tail), not the old stream itself: the caller may have
// destroyed that stream since (its work done), and the runtime does not survive a call on a destroyed stream handle
// (tools/stream_switch_probe.py: a hipEventRecord there takes the process down); an event outlives its stream.
static int mark_tail(fd_handle h, hipStream_t s)
{
    if (!h->ev_switch) FD_HIP(h, hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming));
    FD_HIP(h, hipEventRecord(h->ev_switch, s));
    h->tail_marked = true;
    return FD_OK;
}

static int follow_stream(fd_handle h, hipStream_t s)
{
    if (h->have_last_stream && h->last_stream != s) {
        const int rc = fd_settle(h);      // (a pending check may redo its call on the old stream: that stream must live until the call is settled)
        if (rc != FD_OK) return rc;
        if (h->tail_marked) FD_HIP(h, hipStreamWaitEvent(s, h->ev_switch, 0));
    }
    h->last_stream = s;
    h->have_last_stream = true;
    return FD_OK;
}

// `lens` (host, nullable): valid frames per utterance of a zero-padded batch, uploaded for the kernels from the pinned area `staged`.
// reps: the hoisted predictor's batch holds every utterance once per reverse step (entry n * B + b): its lengths are staged that often
static int set_lens(fd_handle h, const int *lens, int B, int T, hipStream_t stream, const char *who, int *staged, int reps = 1)
{
    h->step_lens = nullptr;
    if (!lens) return FD_OK;
    bool ragged = false;
    for (int b = 0; b < B; ++b) {
        if (lens[b] < 1 || lens[b] > T) FD_FAIL(h, FD_ERR_INVALID, "%s: lens[%d] = %d outside [1, T=%d]", who, b, lens[b], T);
        ragged = ragged || lens[b] < T;
    }
    if (!ragged) return FD_OK;                       // every utterance fills the batch: same launches as without lens
    for (int i = ST_FIRST; i < ST_COUNT; ++i)      // (the step embedding has no time axis)
        if (!h->fast[i])
            FD_FAIL(h, FD_ERR_UNSUPPORTED, "%s: a ragged batch (lens) needs the fast kernel set; the naive kernels (option kernels.<stage> = naive) "
                                           "compute the padded tensor and would silently ignore the lengths", who);
    for (int n = 0; n < reps; ++n) memcpy(staged + (size_t)n * B, lens, sizeof(int) * B);
    FD_HIP(h, hipMemcpyAsync(h->ws.lens_dev, staged, sizeof(int) * B * reps, hipMemcpyHostToDevice, stream));
    h->step_lens = h->ws.lens_dev;
    return FD_OK;
}

int fd_forward(fd_handle h, const float *x, const float *mel, const float *steps, int B, int T, const int *lens,
               float *eps_out, void *stream)
{
    if (h) h->noise_ids.clear();                 // stream ids are for the next fd_sample only: a forward in between drops them
    int rc = check_common(h, B, T, "fd_forward");
    if (rc != FD_OK) return rc;
    if ((rc = fd_settle(h)) != FD_OK) return rc;
    if ((rc = follow_stream(h, (hipStream_t)stream)) != FD_OK) return rc;
    h->inline_fallback = true; h->fp32_mask = 0;      // a single forward always carries its fallbacks inline
    h->hoist_np = 1; h->hoist_step = 0; h->hoist_chunk = false;
    h->embed_valid = false;                      // fd_forward writes its own rows into the same table
    if (!x || !mel || !steps || !eps_out) FD_FAIL(h, FD_ERR_INVALID, "fd_forward: null pointer");
    if (x == eps_out) FD_FAIL(h, FD_ERR_INVALID, "fd_forward: eps_out must not alias x");
    if (h->gen) {
        for (int b = 0; lens && b < B; ++b)
            if (lens[b] < 1 || lens[b] > T) FD_FAIL(h, FD_ERR_INVALID, "fd_forward: lens[%d] = %d outside [1, T=%d]", b, lens[b], T);
        if ((rc = fdg::forward(h, x, mel, steps, B, T, lens, eps_out, (hipStream_t)stream)) != FD_OK) return rc;
        h->last_B = B; h->last_T = T;
        return mark_tail(h, (hipStream_t)stream);
    }
    if ((rc = ensure_workspace(h, B, T)) != FD_OK) return rc;
    if (lens) {
        fd_context::StageSlot *sl = nullptr;
        if ((rc = fd_stage_acquire(h, sizeof(int) * B, &sl)) != FD_OK) return rc;
        if ((rc = set_lens(h, lens, B, T, (hipStream_t)stream, "fd_forward", reinterpret_cast<int *>(sl->host))) != FD_OK) return rc;
        if ((rc = fd_stage_commit(h, sl, (hipStream_t)stream)) != FD_OK) return rc;
    } else {
        h->step_lens = nullptr;
    }
    fdk::Launch L = {h, (hipStream_t)stream, false};
    StepIO io = {x, mel, steps, eps_out, 0};
    hipError_t e = fdk::embed(L, io, B, 1);
    if (e == hipSuccess) e = fdk::clear_range_flags(L);
    if (e == hipSuccess) e = fdk::run_step(L, io, B, T);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_forward: kernel launch failed: %s", hipGetErrorString(e));
    h->last_B = B; h->last_T = T;
    return mark_tail(h, (hipStream_t)stream);
}

static unsigned mode_signature(const fd_context *h)
{
    unsigned s = (h->keep_taps ? 1u : 0u) | (h->gemm_f16 ? 2u : 0u) | (h->lvc_f16 ? 4u : 0u) | (h->conv_f16 ? 8u : 0u) | (h->step_lens ? 16u : 0u) |
                 (h->inline_fallback ? 32u : 0u) | (h->lvc_h8_mfma ? 64u : 0u) | ((unsigned)h->hoist_np << 11) | (h->hoist_chunk ? (1u << 21) : 0u) | (h->fuse_up ? (1u << 22) : 0u) | (h->fuse_advance ? (1u << 23) : 0u) | (h->gemm_wino ? (1u << 24) : 0u);
    for (int i = 0; i < ST_COUNT; ++i) s = (s << 1) | (h->fast[i] ? 1u : 0u);
    return s ^ (h->fp32_mask * 2654435761u);
}

static int resolve_pending(fd_handle h, unsigned *mask);
// Every entry point that touches device state first settles a pending fallback = host check (no-op otherwise).
int fd_settle(fd_handle h)
{
    if (!h->pending.active) return FD_OK;
    unsigned mask = 0;
    const int rc = resolve_pending(h, &mask);
    return rc < 0 ? rc : FD_OK;
}

// `count` consecutive denoiser steps of the current call on `stream`, starting at the device step counter: replayed from captured
// graphs of up to 8 steps (kept per (B, T, mode): there are ~9 us between two graph launches, so a short schedule is one launch
// per call, a long one a series of 8-step launches and a shorter one for the remainder), or launched one by one (options graph = 0,
// profile = 1).  fp32_mask / inline_fallback: fd_internal.h (fd_pipe).
static int enqueue_steps(fd_handle h, int B, int T, int count, unsigned fp32_mask, bool inline_fallback, hipStream_t stream)
{
    Workspace &ws = h->ws;
    h->fp32_mask = fp32_mask;
    h->inline_fallback = inline_fallback;
    StepIO io = {ws.x, ws.mel, nullptr, nullptr, 1};
    struct Restore { fd_handle h; ~Restore() { h->fp32_mask = 0; h->inline_fallback = true; h->hoist_step = 0; if (h->hoist_chunk) h->hoist_np = 1; } } restore{h};
    constexpr int CHUNK = 8;
    // hoist_chunk: the predictor of an np-step piece in front of it, over np * B (step, utterance) entries; the steps then skip theirs
    auto piece_predictor = [&](const fdk::Launch &L, int np) -> hipError_t {
        h->hoist_np = np;
        if (np < 2) return hipSuccess;
        StepIO iop = {ws.x, ws.mel_rep, nullptr, nullptr, np};
        hipError_t e = fdk::kp_front(L, iop, B * np, T);
        return e == hipSuccess ? fdk::kp_gemm(L, B * np, T) : e;
    };
    if (h->hoist_chunk) h->hoist_np = 1;          // (the signature below must not depend on the piece that ran last)
    // between two steps of one sequence the bookkeeping rides in the next step's first kernel -- when that kernel is the fast one
    const bool defer_advance = h->fuse_advance && h->fast[ST_FIRST];
    h->advance_pending = false;
    if (!(h->use_graph && !h->profile)) {
        fdk::Launch L = {h, stream, false};
        for (int k = 0; k < count; ++k) {
            if (h->hoist_chunk && k % CHUNK == 0) {
                hipError_t e = piece_predictor(L, std::min(CHUNK, count - k));
                if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: predictor launch failed: %s", hipGetErrorString(e));
            }
            h->hoist_step = h->hoist_np > 1 ? (h->hoist_chunk ? k % CHUNK : k) : 0;
            hipError_t e = fdk::run_step(L, io, B, T);
            if (e == hipSuccess) {
                // the next step's first kernel does it -- unless a piece predictor runs in between: its front reads the embedding rows
                // through the device step counter, which must already stand at the piece's first step
                if (k + 1 < count && defer_advance && !(h->hoist_chunk && (k + 1) % CHUNK == 0)) h->advance_pending = true;
                else e = fdk::advance_step(L);
            }
            if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: step %d failed: %s", k, hipGetErrorString(e));
        }
        return FD_OK;
    }
    const unsigned sig = mode_signature(h);
    auto graph_of = [&](int steps, hipGraphExec_t *out) -> int {
        for (auto &g : h->graphs)
            if (g.B == B && g.T == T && g.sig == sig && g.steps == steps) {
                g.last_use = ++h->graph_clock;
                *out = g.exec;
                ++h->n_graph_hits;
                return FD_OK;
            }
        reap_retired(h, false);
        while (h->graphs.size() >= (size_t)std::max(1, h->max_graphs)) {
            // evict the least recently used one.  It may still be running (or be queued behind the work on `stream`): retired with an
            // event recorded here, destroyed by a later call once that event has completed -- no wait on this path
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i)
                if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
            fd_context::RetiredGraph r = {h->graphs[lru].graph, h->graphs[lru].exec, nullptr};
            FD_HIP(h, hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
            FD_HIP(h, hipEventRecord(r.done, stream));
            h->retired.push_back(r);
            h->graphs.erase(h->graphs.begin() + lru);
            ++h->n_graph_evictions;
        }
        ++h->n_graph_captures;
        FD_HIP(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        fdk::Launch Lc = {h, h->cap_stream, true};
        hipError_t ec = h->hoist_chunk ? piece_predictor(Lc, steps) : hipSuccess;
        for (int k = 0; k < steps && ec == hipSuccess; ++k) {
            h->hoist_step = h->hoist_np > 1 ? k : 0;      // (hoisted: the graph holds the whole call or piece, so k is its step)
            ec = fdk::run_step(Lc, io, B, T);
            if (ec == hipSuccess) {
                if (k + 1 < steps && defer_advance) h->advance_pending = true;
                else ec = fdk::advance_step(Lc);
            }
        }
        hipGraph_t g = nullptr;
        hipError_t e2 = hipStreamEndCapture(h->cap_stream, &g);
        if (ec != hipSuccess || e2 != hipSuccess) {
            if (g) hipGraphDestroy(g);
            FD_FAIL(h, FD_ERR_HIP, "fd_sample: graph capture failed: %s", hipGetErrorString(ec != hipSuccess ? ec : e2));
        }
        hipGraphExec_t ex = nullptr;
        hipError_t e3 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        if (e3 != hipSuccess) {
            hipGraphDestroy(g);
            FD_FAIL(h, FD_ERR_HIP, "fd_sample: hipGraphInstantiate: %s", hipGetErrorString(e3));
        }
        h->graphs.push_back({B, T, steps, sig, g, ex, ++h->graph_clock});
        *out = ex;
        return FD_OK;
    };
    int rc;
    hipGraphExec_t g_chunk = nullptr, g_rest = nullptr;
    if (count >= CHUNK && (rc = graph_of(CHUNK, &g_chunk)) != FD_OK) return rc;
    if (count % CHUNK && (rc = graph_of(count % CHUNK, &g_rest)) != FD_OK) return rc;
    if (count >= CHUNK && (count % CHUNK)) {      // the second capture may have evicted the first (full cache): look it up again
        if ((rc = graph_of(CHUNK, &g_chunk)) != FD_OK) return rc;
        if ((rc = graph_of(count % CHUNK, &g_rest)) != FD_OK) return rc;
    }
    for (int k = 0; k + CHUNK <= count; k += CHUNK) FD_HIP(h, hipGraphLaunch(g_chunk, stream));
    if (count % CHUNK) FD_HIP(h, hipGraphLaunch(g_rest, stream));
    return FD_OK;
}

static int sample_core(fd_handle h, const fd_context::SampleArgs &a, unsigned force_mask, long long ticket);

// Frames of the library's own buffers for a sample call of T frames (fd_context::t_bucket): T rounded up to the bucket when the call
// is replayed from graphs and every stage runs the kernel set that honours `lens`; T itself otherwise.
static int bucket_frames(const fd_context *h, int T)
{
    if (h->t_bucket <= 1 || !h->use_graph || h->profile || h->keep_taps) return T;
    for (int i = ST_FIRST; i < ST_COUNT; ++i)
        if (!h->fast[i]) return T;
    const int64_t tp = ((int64_t)T + h->t_bucket - 1) / h->t_bucket * h->t_bucket;
    return tp > 0x3fffffff ? T : (int)tp;
}

// How many reverse steps' kernels one predictor launch pair computes for this call: N (hoisted) or 1 (the predictor stays in the step)
static int hoist_mult(const fd_context *h, int B, int T, int N)
{
    if (h->hoist_mode == 0 || N < 2) return 1;
    if (!(h->fast[ST_KP_FRONT] && h->fast[ST_KP_GEMM] && h->fast[ST_LVC]) || h->keep_taps) return 1;
    const int np = std::min(N, 8);             // a longer schedule hoists per 8-step graph piece (fd_context::hoist_chunk)
    if (h->hoist_mode == 2) return np;
    return (int64_t)B * T <= 4096 ? np : 1;        // measured at T = 864: B = 1 -7.8 %, 2 -6.2 %, 3 -4.1 %, 4 -1.9 %, 8 and 16 +-0 (profiles/r03/s20_*)
}

// fallback = host: waits for the pending piece of work, looks at its range flags and, if one was raised, runs that piece again with
// the flagged stages on their fp32 kernels (and every other stage with its fallback inline: the second pass is always right) -- a
// lazily checked call (<= 8 steps) as a whole from its arguments, a piece of a long schedule from the saved x.
// Returns 1 if it redid the work, 0 if not; *mask receives the flagged stages (sticky for the rest of a long call).
static int resolve_call(fd_handle h, const fd_context::PendingCall &p, unsigned *mask)
{
    FD_HIP(h, hipEventSynchronize(p.slot ? h->flags_done2 : h->flags_done));
    const int *fl = h->flags_host + 32 * p.slot;
    unsigned m = 0;
    for (int i = 0; i < 32; ++i)
        if (fl[i]) m |= 1u << i;
    if (m & (1u << 19)) m |= 1u;                 // the predictor front feeds the GEMM: both go
    *mask |= m;
    if (m == 0) return 0;
    h->redone_ring[h->redone_next++ % 16] = p.ticket;
    if (p.lazy) ++h->n_calls_redone;
    else { ++h->n_pieces_redone; h->call_fp32_mask |= *mask; }
    if (p.lazy) {
        const int rc = sample_core(h, p.args, *mask, p.ticket);
        return rc < 0 ? rc : 1;
    }
    Workspace &ws = h->ws;
    const size_t n_el = (size_t)p.B * p.T * fd::HOPT;
    FD_HIP(h, hipMemcpyAsync(ws.x, ws.xsave, sizeof(float) * n_el, hipMemcpyDeviceToDevice, p.stream));
    fdk::Launch L = {h, p.stream, false};
    hipError_t e = fdk::clear_range_flags(L, p.first);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample_check: %s", hipGetErrorString(e));
    int rc = enqueue_steps(h, p.B, p.T, p.count, *mask, true, p.stream);
    if (rc != FD_OK) return rc;
    if (p.first + p.count == p.N) {
        e = fdk::copy_rows(L, p.out, (int64_t)p.T_io * fd::HOPT, ws.x, (int64_t)p.T * fd::HOPT, p.T_io * fd::HOPT, p.B);
        if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample_check: %s", hipGetErrorString(e));
    }
    return 1;
}

static int resolve_pending(fd_handle h, unsigned *mask)
{
    fd_context::PendingCall p = h->pending;
    h->pending.active = false;
    return resolve_call(h, p, mask);
}

// One whole sample call on a.stream.  force_mask != 0: a redo (the flagged stages on fp32, every other fallback inline, nothing left
// pending); otherwise the handle's mode: in-graph fallbacks, or fallback = host (lazy for N <= 8, checked every 8 steps beyond).
static int sample_core(fd_handle h, const fd_context::SampleArgs &a, unsigned force_mask, long long ticket)
{
    int rc;
    // T: frames of the library's buffers (the caller's T_io rounded up to the bucket); the caller's tensors keep their dense T_io layout
    const int B = a.B, T_io = a.T, T = bucket_frames(h, a.T), N = a.N;
    const int64_t L_io = (int64_t)T_io * fd::HOPT, Lp = (int64_t)T * fd::HOPT;
    hipStream_t stream = a.stream;
    Workspace &ws = h->ws;
    const size_t n_el = (size_t)B * T * fd::HOPT;
    const std::vector<unsigned long long> &ids = a.ids;
    std::vector<int> own_lens;                  // a bucketed call without `lens`: every utterance is T_io of the T frames long
    const int *lens_eff = a.has_lens ? a.lens.data() : nullptr;
    if (!lens_eff && T != T_io) { own_lens.assign(B, T_io); lens_eff = own_lens.data(); }
    // per-call parameters -> device block the captured kernels read.  Staged through the pinned ring: the call returns
    // without waiting for the stream, so the host prepares the next call while this one runs.
    {
        fd_context::StageSlot *sl = nullptr;
        const int np = hoist_mult(h, B, T, N);      // the hoisted predictor's batch: np reverse steps x B utterances (below)
        const size_t off_lens = sizeof(StepParams), off_ids = off_lens + ((sizeof(int) * B * np + 7) & ~(size_t)7);
        if ((rc = fd_stage_acquire(h, off_ids + sizeof(unsigned long long) * B, &sl)) != FD_OK) return rc;
        if ((rc = set_lens(h, lens_eff, B, T, stream, "fd_sample", reinterpret_cast<int *>(sl->host + off_lens), np)) != FD_OK) return rc;
        if (!ids.empty()) {
            memcpy(sl->host + off_ids, ids.data(), sizeof(unsigned long long) * B);
            FD_HIP(h, hipMemcpyAsync(ws.uid_dev, sl->host + off_ids, sizeof(unsigned long long) * B, hipMemcpyHostToDevice, stream));
        }
        StepParams *p = reinterpret_cast<StepParams *>(sl->host);
        memcpy(p->table, a.table.data(), sizeof(fd_step) * N);
        p->z = a.z; p->seq = a.seq_out; p->seed = a.seed; p->n_steps = N; p->ddim = a.ddim ? 1 : 0; p->step_idx = 0; p->l4 = T * (fd::HOPT / 4);
        p->uids = ids.empty() ? nullptr : ws.uid_dev;
        p->l4_io = T_io * (fd::HOPT / 4); p->n4_io = (long long)B * p->l4_io;
        // only the used prefix of the table plus the trailer needs to travel
        const size_t head = sizeof(fd_step) * N;
        FD_HIP(h, hipMemcpyAsync(ws.params, p, head, hipMemcpyHostToDevice, stream));
        const size_t off = offsetof(StepParams, z);
        FD_HIP(h, hipMemcpyAsync(reinterpret_cast<char *>(ws.params) + off, reinterpret_cast<const char *>(p) + off, sizeof(StepParams) - off,
                                 hipMemcpyHostToDevice, stream));
        if ((rc = fd_stage_commit(h, sl, stream)) != FD_OK) return rc;
    }
    fdk::Launch L = {h, stream, false};
    hipError_t e = fdk::copy_rows(L, ws.mel, T, a.mel, T_io, T_io, B * fd::COND);
    if (e == hipSuccess && a.x_T) e = fdk::copy_rows(L, ws.x, Lp, a.x_T, L_io, (int)L_io, B);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: input copy failed: %s", hipGetErrorString(e));
    if (!a.x_T && (e = fdk::init_noise(L, ws.x, B, T * (fd::HOPT / 4), T_io * (fd::HOPT / 4), a.seed, ids.empty() ? nullptr : ws.uid_dev)) != hipSuccess)
        FD_FAIL(h, FD_ERR_HIP, "fd_sample: init_noise failed: %s", hipGetErrorString(e));
    if (a.seq_out && (e = fdk::copy_rows(L, a.seq_out, L_io, ws.x, Lp, (int)L_io, B)) != hipSuccess)
        FD_FAIL(h, FD_ERR_HIP, "fd_sample: sequence copy failed: %s", hipGetErrorString(e));
    // the waveform back into the caller's dense tensor
    auto copy_out = [&]() -> int {
        const hipError_t eo = fdk::copy_rows(L, a.out, L_io, ws.x, Lp, (int)L_io, B);
        if (eo != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: output copy failed: %s", hipGetErrorString(eo));
        return FD_OK;
    };

    StepIO io = {ws.x, ws.mel, nullptr, nullptr, 1};
    {   // the embedding rows of this schedule: still in ws.noise from the previous call?
        std::vector<float> ts(N);
        for (int k = 0; k < N; ++k) ts[k] = a.table[k].t;
        if (!(h->embed_cache && h->embed_valid && h->embed_B == B && h->embed_t == ts)) {
            h->embed_valid = false;
            if ((e = fdk::embed(L, io, B, N)) != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: embed failed: %s", hipGetErrorString(e));
            h->embed_t.swap(ts);
            h->embed_B = B;
            h->embed_valid = true;
        }
    }
    if ((e = fdk::clear_range_flags(L)) != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: %s", hipGetErrorString(e));

    constexpr int CHUNK = 8;
    // Hoisted predictor (fd_internal.h: hoist_np): one front + GEMM launch pair over the batch of N * B (step, utterance) entries
    h->hoist_np = hoist_mult(h, B, T, N);
    h->hoist_step = 0;
    h->hoist_chunk = h->hoist_np > 1 && N > CHUNK;
    if (h->hoist_np > 1) {      // the mel once per predicted step: ONE launch (the lengths were staged that often by set_lens)
        const int64_t mel_n = (int64_t)B * fd::COND * T;
        if ((e = fdk::copy_rows(L, ws.mel_rep, T, ws.mel, T, T, B * fd::COND, h->hoist_np, mel_n)) != hipSuccess)
            FD_FAIL(h, FD_ERR_HIP, "fd_sample: mel replication failed: %s", hipGetErrorString(e));
    }
    if (h->hoist_chunk) h->hoist_np = 1;          // enqueue_steps sets it piece by piece
    else if (h->hoist_np > 1) {
        h->fp32_mask = force_mask;
        h->inline_fallback = force_mask != 0 || !h->host_fallback;
        StepIO iop = {ws.x, ws.mel_rep, nullptr, nullptr, 0};      // "forward" addressing: batch entry n * B + b reads noise row n * B + b
        e = fdk::kp_front(L, iop, B * N, T);
        if (e == hipSuccess) e = fdk::kp_gemm(L, B * N, T);
        h->fp32_mask = 0; h->inline_fallback = true;
        if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_sample: predictor launch failed: %s", hipGetErrorString(e));
    }
    if (force_mask != 0) {
        if ((rc = enqueue_steps(h, B, T, N, force_mask, true, stream)) != FD_OK) return rc;
        if ((rc = copy_out()) != FD_OK) return rc;
    } else if (h->host_fallback && N <= CHUNK) {
        // fallback = host, one graph launch: no fp32 launch trails the fp16x2 kernels; their flags accumulate on the device and travel
        // to the host behind the work.  Looked at lazily: by the next fd_sample after it has enqueued itself, or by fd_sample_check /
        // fd_sample_settle; a flagged call is then run again as a whole.
        if ((rc = enqueue_steps(h, B, T, N, 0u, /*inline_fallback=*/false, stream)) != FD_OK) return rc;
        const int slot = (int)(ticket & 1);
        FD_HIP(h, hipMemcpyAsync(h->flags_host + 32 * slot, ws.range_flag + 64, sizeof(int) * 32, hipMemcpyDeviceToHost, stream));
        FD_HIP(h, hipEventRecord(slot ? h->flags_done2 : h->flags_done, stream));
        if ((rc = copy_out()) != FD_OK) return rc;      // provisional until checked
        h->pending.active = true; h->pending.lazy = true; h->pending.slot = slot; h->pending.ticket = ticket;
        h->pending.B = B; h->pending.T = T; h->pending.T_io = T_io; h->pending.N = N; h->pending.first = 0; h->pending.count = N; h->pending.out = a.out;
        h->pending.stream = stream; h->pending.args = a;
    } else if (h->host_fallback) {
        // a long schedule: each 8-step piece is checked (one stream synchronisation) before the next is enqueued, and redone from the
        // saved x with the flagged stages on their fp32 kernels once the host has seen them
        FD_HIP(h, hipMemcpyAsync(ws.xsave, ws.x, sizeof(float) * n_el, hipMemcpyDeviceToDevice, stream));
        unsigned mask = 0;
        for (int first = 0; first < N; first += CHUNK) {
            const int count = std::min(CHUNK, N - first);
            const bool last = first + count == N;
            ++h->n_pieces;
            if (mask != 0) ++h->n_pieces_fp32;
            if (first > 0 && mask == 0) FD_HIP(h, hipMemcpyAsync(ws.xsave, ws.x, sizeof(float) * n_el, hipMemcpyDeviceToDevice, stream));
            if ((rc = enqueue_steps(h, B, T, count, mask, /*inline_fallback=*/mask != 0, stream)) != FD_OK) return rc;
            if (mask != 0) continue;              // already on the safe path: nothing to look at
            FD_HIP(h, hipMemcpyAsync(h->flags_host, ws.range_flag + 64, sizeof(int) * 32, hipMemcpyDeviceToHost, stream));
            FD_HIP(h, hipEventRecord(h->flags_done, stream));
            h->pending.active = true; h->pending.lazy = false; h->pending.slot = 0; h->pending.ticket = ticket;
            h->pending.B = B; h->pending.T = T; h->pending.T_io = T_io; h->pending.N = N; h->pending.first = first; h->pending.count = count; h->pending.out = a.out;
            h->pending.stream = stream;
            if (last) break;                      // the caller's fd_sample_check (or the next call on this handle) looks at it
            int redone = resolve_pending(h, &mask);
            if (redone < 0) return redone;
        }
        if ((rc = copy_out()) != FD_OK) return rc;      // (provisional while a check is pending)
    } else {
        if ((rc = enqueue_steps(h, B, T, N, 0u, true, stream)) != FD_OK) return rc;
        if ((rc = copy_out()) != FD_OK) return rc;
    }
    h->last_B = B; h->last_T = T;
    return mark_tail(h, stream);      // (also behind a redo: resolve_call comes through here)
}

int fd_sample(fd_handle h, const float *mel, int B, int T, const int *lens, const fd_step *table, int N, int ddim,
              const float *x_T, const float *z, uint64_t seed, float *out, float *seq_out, void *stream_)
{
    std::vector<unsigned long long> ids;
    if (h) ids.swap(h->noise_ids);               // one-shot: fd_set_noise_streams applies to this call only, also when it fails below
    int rc = check_common(h, B, T, "fd_sample");
    if (rc != FD_OK) return rc;
    if ((rc = follow_stream(h, (hipStream_t)stream_)) != FD_OK) return rc;
    if (h->gen) {      // a configuration other than base.yaml's: exact-fp32 kernels, nothing provisional, no graph
        if (!mel || !table || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_sample: null pointer");
        if (N <= 0 || N > 1024) FD_FAIL(h, FD_ERR_INVALID, "fd_sample: N=%d outside 1..1024", N);
        if (!ids.empty() && (int)ids.size() != B) FD_FAIL(h, FD_ERR_INVALID, "fd_sample: fd_set_noise_streams gave %d stream ids but B=%d", (int)ids.size(), B);
        for (int b = 0; lens && b < B; ++b)
            if (lens[b] < 1 || lens[b] > T) FD_FAIL(h, FD_ERR_INVALID, "fd_sample: lens[%d] = %d outside [1, T=%d]", b, lens[b], T);
        if ((rc = fdg::sample(h, mel, B, T, lens, table, N, ddim, x_T, z, seed, ids, out, seq_out, (hipStream_t)stream_)) != FD_OK) return rc;
        ++h->ticket_counter;
        h->last_B = B; h->last_T = T;
        return mark_tail(h, (hipStream_t)stream_);
    }
    // A lazily checked previous call (fallback = host, <= 8 steps) is looked at AFTER this call has enqueued its own work -- unless
    // this call cannot be lazy itself, or the workspace must grow first (that waits for the device anyway).
    const bool lazy = h->host_fallback && N >= 1 && N <= 8;
    const int Tp = bucket_frames(h, T);          // frames of the library's own buffers (t_bucket): what the workspace and the graphs are sized for
    if ((rc = check_common(h, B, Tp, "fd_sample")) != FD_OK) return rc;
    const int np = (N >= 1 && N <= 1024) ? hoist_mult(h, B, Tp, N) : 1;
    const bool ws_ok = workspace_fits(h, B, Tp, np);
    fd_context::PendingCall prev;
    if (lazy && ws_ok && h->pending.active && h->pending.lazy) {
        prev = h->pending;
        h->pending.active = false;
    } else if ((rc = fd_settle(h)) != FD_OK) return rc;
    auto finish_prev = [&]() -> int {
        if (!prev.active) return FD_OK;
        unsigned mask = 0;
        const fd_context::PendingCall cur = h->pending;      // a redo of the previous call must not disturb this call's record
        h->pending.active = false;
        const int r = resolve_call(h, prev, &mask);
        h->pending = cur;
        prev.active = false;
        return r < 0 ? r : FD_OK;
    };
    if (!mel || !table || !out) { finish_prev(); FD_FAIL(h, FD_ERR_INVALID, "fd_sample: null pointer"); }
    if (N <= 0 || N > 1024) { finish_prev(); FD_FAIL(h, FD_ERR_INVALID, "fd_sample: N=%d outside 1..1024", N); }
    if (!ids.empty() && (int)ids.size() != B) {
        finish_prev();
        FD_FAIL(h, FD_ERR_INVALID, "fd_sample: fd_set_noise_streams gave %d stream ids but B=%d", (int)ids.size(), B);
    }
    for (int b = 0; lens && b < B; ++b)
        if (lens[b] < 1 || lens[b] > T) {
            finish_prev();
            FD_FAIL(h, FD_ERR_INVALID, "fd_sample: lens[%d] = %d outside [1, T=%d]", b, lens[b], T);
        }
    if ((rc = ensure_workspace(h, B, Tp, np)) != FD_OK) { finish_prev(); return rc; }
    fd_context::SampleArgs a;
    a.mel = mel; a.B = B; a.T = T; a.N = N; a.ddim = ddim;
    a.has_lens = lens != nullptr;
    if (lens) a.lens.assign(lens, lens + B);
    a.table.assign(table, table + N);
    a.x_T = x_T; a.z = z; a.seed = seed; a.out = out; a.seq_out = seq_out; a.stream = (hipStream_t)stream_;
    a.ids.swap(ids);
    const long long ticket = ++h->ticket_counter;
    h->n_pieces = h->n_pieces_redone = h->n_pieces_fp32 = 0;
    h->call_fp32_mask = 0;
    rc = sample_core(h, a, 0u, ticket);
    const int rc_prev = finish_prev();
    h->last_B = B; h->last_T = Tp;               // (a redo of the previous call has just run with that call's shape)
    if (rc != FD_OK || rc_prev != FD_OK) return rc != FD_OK ? rc : rc_prev;
    // The contract of the reference call is "call, then read" (util.py:215-235): unless the caller opted into the pipelined check
    // (option defer_check = 1: tickets, fd_sample_check / fd_sample_settle), the range check of this call is settled before fd_sample
    // returns -- one wait for the call's own work and, if an operand left the fp16 range, the second pass on the fp32 kernels.
    if (!h->defer_check) return fd_settle(h);
    return FD_OK;
}

int64_t fd_sample_ticket(fd_handle h) { return h ? (int64_t)h->ticket_counter : FD_ERR_INVALID; }

// 1 if call `ticket` had to be redone on the fp32 kernels (now, or when a later fd_sample looked at it), 0 if not.
static int ticket_redone(fd_handle h, long long ticket)
{
    for (long long t : h->redone_ring)
        if (t == ticket && t != 0) return 1;
    return 0;
}

int fd_sample_settle(fd_handle h, int64_t ticket)
{
    if (!h) return FD_ERR_INVALID;
    if (ticket <= 0 || ticket > h->ticket_counter) FD_FAIL(h, FD_ERR_INVALID, "fd_sample_settle: unknown ticket %lld", (long long)ticket);
    if (h->pending.active && h->pending.ticket <= ticket) {
        FD_HIP(h, hipSetDevice(h->device));
        unsigned mask = 0;
        const int r = resolve_pending(h, &mask);
        if (r < 0) return r;
    }
    return ticket_redone(h, ticket);
}

int fd_sample_check(fd_handle h)
{
    if (!h) return FD_ERR_INVALID;
    if (!h->pending.active) return 0;
    FD_HIP(h, hipSetDevice(h->device));
    unsigned mask = 0;
    return resolve_pending(h, &mask);
}

int fd_set_noise_streams(fd_handle h, const uint64_t *stream_ids, int B)
{
    if (!h || B < 0 || (B > 0 && !stream_ids)) return FD_ERR_INVALID;
    h->noise_ids.assign(stream_ids, stream_ids + B);
    return FD_OK;
}

// ------------------------------------------------------------------------------------------------
// options, taps, profile
// ------------------------------------------------------------------------------------------------
int fd_set_option(fd_handle h, const char *key, const char *value)
{
    if (!h || !key || !value) return FD_ERR_INVALID;
    {
        const int rcs = fd_settle(h);
        if (rcs != FD_OK) return rcs;
    }
    const std::string k(key), v(value);
    static const char *stage_names[ST_COUNT] = {"embed", "first", "dblock", "kp_front", "kp_gemm", "convt", "lvc", "final"};
    auto parse_mode = [&](bool &dst) -> int {
        if (v == "fast") dst = true;
        else if (v == "naive") dst = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: %s expects fast|naive, got '%s'", key, value);
        return FD_OK;
    };
    if (k == "kernels") {
        bool m = true;
        int rc = parse_mode(m);
        if (rc != FD_OK) return rc;
        for (int i = 0; i < ST_COUNT; ++i) h->fast[i] = m;
        return FD_OK;
    }
    if (k.compare(0, 8, "kernels.") == 0) {
        for (int i = 0; i < ST_COUNT; ++i)
            if (k.substr(8) == stage_names[i]) return parse_mode(h->fast[i]);
        FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: unknown stage '%s'", key);
    }
    const bool on = (v == "1" || v == "true" || v == "on");
    if (k == "gemm") {
        if (v == "f16x2") h->gemm_f16 = true;
        else if (v == "fp32") h->gemm_f16 = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: gemm expects f16x2|fp32, got '%s'", value);
        return FD_OK;
    }
    if (k == "gemm_form") {      // how the fp16x2 predictor GEMM evaluates kernel_conv's three taps
        if (v == "winograd") h->gemm_wino = true;
        else if (v == "direct") h->gemm_wino = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: gemm_form expects winograd|direct, got '%s'", value);
        return FD_OK;
    }
    if (k == "lvc") {
        if (v == "f16x2") h->lvc_f16 = true;
        else if (v == "fp32") h->lvc_f16 = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: lvc expects f16x2|fp32, got '%s'", value);
        return FD_OK;
    }
    if (k == "conv") {
        if (v == "f16x2") h->conv_f16 = true;
        else if (v == "fp32") h->conv_f16 = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: conv expects f16x2|fp32, got '%s'", value);
        return FD_OK;
    }
    if (k == "mel") {
        if (v == "pwg") h->mel_variant = MEL_PWG;
        else if (v == "tacotron") h->mel_variant = MEL_TACOTRON;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: mel expects pwg|tacotron, got '%s'", value);
        return FD_OK;
    }
    if (k == "lvc_h8") {
        if (v == "mfma") h->lvc_h8_mfma = true;
        else if (v == "valu") h->lvc_h8_mfma = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: lvc_h8 expects mfma|valu, got '%s'", value);
        drop_graph(h);
        return FD_OK;
    }
    if (k == "fallback") {
        if (v == "host") h->host_fallback = true;
        else if (v == "graph") h->host_fallback = false;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: fallback expects graph|host, got '%s'", value);
        return FD_OK;
    }
    if (k == "fuse_final") { h->fuse_final = on; drop_graph(h); return FD_OK; }
    if (k == "fuse_up") { h->fuse_up = on; drop_graph(h); return FD_OK; }
    if (k == "fuse_advance") { h->fuse_advance = on; drop_graph(h); return FD_OK; }
    if (k == "lvc_dx") {      // training operator, frames path: gather = dx reads the forward-order frames; copy = a reordered copy first
        if (v != "gather" && v != "copy") FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: lvc_dx expects gather|copy, got '%s'", value);
        h->lvc_dx_gather = (v == "gather"); return FD_OK;
    }
    if (k == "embed_cache") { h->embed_cache = on; return FD_OK; }
    if (k == "hoist") {
        if (v == "auto") h->hoist_mode = 1;
        else if (v == "on") h->hoist_mode = 2;
        else if (v == "off") h->hoist_mode = 0;
        else FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: hoist expects auto|on|off, got '%s'", value);
        return FD_OK;
    }
    if (k == "graph") { h->use_graph = on; return FD_OK; }
    if (k == "defer_check") { h->defer_check = on; return FD_OK; }
    if (k == "t_bucket" || k == "graph_cache") {
        char *end = nullptr;
        const long n = strtol(value, &end, 10);
        if (end == value || *end != 0 || n < 0 || n > 65536 || (k == "graph_cache" && n < 1))
            FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: %s expects an integer (t_bucket: frames, 0 = exact T; graph_cache: graphs kept, >= 1), got '%s'", key, value);
        if (k == "t_bucket") h->t_bucket = (int)n;
        else h->max_graphs = (int)n;
        return FD_OK;
    }
    if (k == "profile") { h->profile = (v == "events") ? 2 : (on ? 1 : 0); return FD_OK; }
    if (k == "taps") { h->keep_taps = on; return FD_OK; }
    FD_FAIL(h, FD_ERR_INVALID, "fd_set_option: unknown option '%s'", key);
}

}  // extern "C"
