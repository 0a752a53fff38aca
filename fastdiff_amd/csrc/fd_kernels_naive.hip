// fd_kernels_naive.hip -- straightforward one-thread-per-output HIP kernels for every stage of the path.
// They exist to (a) bring the path up, (b) isolate a wrong fast kernel stage by stage on the GPU
// (option "kernels.<stage>"="naive"), never as the shipped configuration.  The embed, update, Philox and
// epilogue kernels at the bottom are shared by both modes.
#include "fd_kernels.h"
#include "fd_device.h"

namespace fdk_naive {

// out[b,o,t] = post(bias[o] + sum_{i,k} w[o,i,k] * pre(x[b,i,(t + k*dil - pad)*in_stride] + add_in[b,i])) (+ residual)
struct ConvArgs {
    const float *x, *w, *b, *residual, *add_in;
    float *out;
    const int *perm;
    int B, Cin, Cout, L, KS, dil, in_stride, in_len;
    float pre_slope, post_slope;
    int out_mode, out_rec, out_off;   // out_mode 1: out[(b*L + t)*out_rec + out_off + perm[o]]
    int add_in_stride;                // stride between batch rows of add_in
    const int *step_idx_ptr;          // sampler mode: add_in += *step_idx_ptr * add_in_step_stride
    int add_in_step_stride;
};

__global__ void k_conv1d(ConvArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.Cout * a.L;
    if (idx >= total) return;
    const int t = (int)(idx % a.L);
    const int o = (int)((idx / a.L) % a.Cout);
    const int b = (int)(idx / ((int64_t)a.L * a.Cout));
    const int pad = a.dil * (a.KS - 1) / 2;
    float acc = a.b[o];
    for (int i = 0; i < a.Cin; ++i) {
        const float *xr = a.x + ((int64_t)b * a.Cin + i) * a.in_len;
        const float *addp = a.add_in;
        if (addp && a.step_idx_ptr) addp += (int64_t)(*a.step_idx_ptr) * a.add_in_step_stride;
        const float addv = addp ? addp[(int64_t)b * a.add_in_stride + i] : 0.0f;
        for (int k = 0; k < a.KS; ++k) {
            const int p = t + k * a.dil - pad;
            if (p < 0 || p >= a.L) continue;
            float v = xr[(int64_t)p * a.in_stride] + addv;
            v = v > 0.0f ? v : v * a.pre_slope;
            acc += a.w[((int64_t)o * a.Cin + i) * a.KS + k] * v;
        }
    }
    acc = acc > 0.0f ? acc : acc * a.post_slope;
    if (a.residual) acc += a.residual[((int64_t)b * a.Cout + o) * a.L + t];
    if (a.out_mode == 0)
        a.out[((int64_t)b * a.Cout + o) * a.L + t] = acc;
    else
        a.out[((int64_t)b * a.L + t) * a.out_rec + a.out_off + (a.perm ? a.perm[o] : o)] = acc;
}

static hipError_t conv1d(const fdk::Launch &L, const char *name, const ConvArgs &a)
{
    const int64_t total = (int64_t)a.B * a.Cout * a.L;
    const int grid = (int)((total + 255) / 256);
    FD_LAUNCH(L, name, k_conv1d, dim3(grid), dim3(256), 0, a);
    return hipSuccess;
}

// ConvTranspose1d(32,32,2r,stride r,pad r/2) of leaky_relu(x,0.2); weight [in][out][2r]
__global__ void k_convt(const float *x, const float *w, const float *bias, float *out, int B, int Lin, int r)
{
    const int Lout = Lin * r;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * fd::C * Lout) return;
    const int t = (int)(idx % Lout);
    const int o = (int)((idx / Lout) % fd::C);
    const int b = (int)(idx / ((int64_t)Lout * fd::C));
    const int p = r / 2;
    float acc = bias[o];
    const int j1 = (t + p) / r;
    for (int j = j1 - 1; j <= j1; ++j) {
        const int k = t + p - j * r;
        if (j < 0 || j >= Lin || k < 0 || k >= 2 * r) continue;
        for (int i = 0; i < fd::C; ++i) {
            float v = x[((int64_t)b * fd::C + i) * Lin + j];
            v = v > 0.0f ? v : 0.2f * v;
            acc += v * w[((int64_t)i * fd::C + o) * 2 * r + k];
        }
    }
    out[idx] = acc;
}

__global__ void k_add_inplace(float *x, const float *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += y[i];
}

// location-variable convolution + gate, in place on x:  x[b,ch,t] += sigmoid(z[ch]) * tanh(z[ch+32])
// z[o] = bias_f[o] + sum_{i,k} ypad[b,i,t+k-1] * K_f[i][o][k],  f = t / hop   (modules.py:217,232-253)
__global__ void k_lvc_gate(const float *y, const float *kpack, float *x, int B, int T, int hop, int layer)
{
    const int Ln = T * hop;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * fd::C * Ln) return;
    const int t = (int)(idx % Ln);
    const int ch = (int)((idx / Ln) % fd::C);
    const int b = (int)(idx / ((int64_t)Ln * fd::C));
    const int f = t / hop;
    const float *rec = kpack + ((int64_t)b * T + f) * fd::KREC;
    float zs = rec[fd::bias_index(layer, ch)];
    float zt = rec[fd::bias_index(layer, ch + fd::C)];
    for (int i = 0; i < fd::C; ++i)
        for (int k = 0; k < 3; ++k) {
            const int p = t + k - 1;
            if (p < 0 || p >= Ln) continue;
            const float v = y[((int64_t)b * fd::C + i) * Ln + p];
            zs += v * rec[fd::kernel_index(layer, i, ch, k)];
            zt += v * rec[fd::kernel_index(layer, i, ch + fd::C, k)];
        }
    const float sg = 1.0f / (1.0f + expf(-zs));
    x[idx] += sg * tanhf(zt);
}

}  // namespace fdk_naive

// ------------------------------------------------------------------------------------------------
// stage drivers (naive mode)
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_naive;

hipError_t naive_first_conv(const Launch &L, const StepIO &io, int B, int T)
{
    const DevWeights &w = L.ctx->w;
    ConvArgs a = {};
    a.x = io.x_in; a.w = w.first.w; a.b = w.first.b; a.out = L.ctx->ws.a[0];
    a.B = B; a.Cin = 1; a.Cout = fd::C; a.L = T * fd::HOPT; a.KS = 7; a.dil = 1; a.in_stride = 1; a.in_len = a.L;
    a.pre_slope = 1.0f; a.post_slope = 1.0f;
    return conv1d(L, "naive_first_conv", a);
}

hipError_t naive_dblock(const Launch &L, int d, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    int Lin = T * fd::HOPT;
    for (int i = 0; i < d; ++i) Lin /= fd::down_factor(i);
    const int f = fd::down_factor(d), Lo = Lin / f;
    float *tmpA = c->ws.xA, *tmpB = c->ws.xB;     // scratch (free during the down path)
    float *res = tmpA + (int64_t)B * fd::C * Lo;  // second half of xA
    hipError_t e;
    ConvArgs a = {};
    a.B = B; a.Cin = fd::C; a.Cout = fd::C; a.L = Lo; a.pre_slope = 1.0f; a.post_slope = 1.0f;
    // residual = interpolate(residual_dense(x)) == 1x1 conv on the strided pick
    a.x = c->ws.a[d]; a.in_stride = f; a.in_len = Lin; a.w = w.down[d].res.w; a.b = w.down[d].res.b; a.KS = 1; a.dil = 1; a.out = res;
    if ((e = conv1d(L, "naive_dblock_res", a)) != hipSuccess) return e;
    // three dilated convs on leaky_relu(.)
    const int dil[3] = {1, 2, 4};
    const float *in = c->ws.a[d];
    int in_stride = f, in_len = Lin;
    float *outs[3] = {tmpA, tmpB, c->ws.a[d + 1]};
    for (int l = 0; l < 3; ++l) {
        a.x = in; a.in_stride = in_stride; a.in_len = in_len; a.w = w.down[d].conv[l].w; a.b = w.down[d].conv[l].b;
        a.KS = 3; a.dil = dil[l]; a.pre_slope = 0.2f; a.out = outs[l]; a.residual = (l == 2) ? res : nullptr;
        if ((e = conv1d(L, "naive_dblock_conv", a)) != hipSuccess) return e;
        in = outs[l]; in_stride = 1; in_len = Lo;
    }
    return hipSuccess;
}

hipError_t naive_kp_front(const Launch &L, const StepIO &io, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    hipError_t e;
    for (int n = 0; n < fd::NBLK; ++n) {
        const int64_t hsz = (int64_t)B * fd::HID * T;
        float *h0 = c->ws.kp_h0 + n * hsz, *hA = c->ws.kp_hA + n * hsz, *hB = c->ws.kp_hB + n * hsz;
        ConvArgs a = {};
        a.B = B; a.L = T; a.in_stride = 1; a.in_len = T; a.dil = 1; a.pre_slope = 1.0f; a.post_slope = 0.1f;
        a.x = io.mel; a.Cin = fd::COND; a.Cout = fd::HID; a.KS = 5; a.w = w.blk[n].kp_in.w; a.b = w.blk[n].kp_in.b; a.out = h0;
        // noise[step][b][blk][80]; in sampler mode the step index lives on the device (captured graph)
        a.add_in = c->ws.noise + n * fd::COND; a.add_in_stride = fd::NBLK * fd::COND;
        a.step_idx_ptr = io.sampler ? &c->ws.params->step_idx : nullptr;
        a.add_in_step_stride = B * fd::NBLK * fd::COND;
        if ((e = conv1d(L, "naive_kp_in", a)) != hipSuccess) return e;
        a.add_in = nullptr; a.step_idx_ptr = nullptr; a.Cin = fd::HID; a.KS = 3;
        const float *in = h0;
        for (int l = 0; l < 6; ++l) {     // h0 -> A -> B -> A -> B -> A -> B (+h0): the result is always in kp_hB
            float *out = (l & 1) ? hB : hA;
            a.x = in; a.w = w.blk[n].kp_res[l].w; a.b = w.blk[n].kp_res[l].b; a.out = out;
            a.residual = (l == 5) ? h0 : nullptr;
            if ((e = conv1d(L, "naive_kp_res", a)) != hipSuccess) return e;
            in = out;
        }
    }
    return hipSuccess;
}

hipError_t naive_kp_gemm(const Launch &L, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    hipError_t e;
    for (int n = 0; n < fd::NBLK; ++n) {
        const int64_t hsz = (int64_t)B * fd::HID * T;
        ConvArgs a = {};
        a.B = B; a.L = T; a.in_stride = 1; a.in_len = T; a.dil = 1; a.pre_slope = 1.0f; a.post_slope = 1.0f;
        a.x = c->ws.kp_hB + n * hsz; a.Cin = fd::HID; a.KS = 3;
        a.out = c->ws.kpack + (int64_t)n * B * T * fd::KREC; a.out_mode = 1; a.out_rec = fd::KREC;
        a.Cout = fd::KW; a.w = w.blk[n].kc.w; a.b = w.blk[n].kc.b; a.perm = w.kc_perm; a.out_off = 0;
        if ((e = conv1d(L, "naive_kernel_conv", a)) != hipSuccess) return e;
        a.Cout = fd::KB; a.w = w.blk[n].bc.w; a.b = w.blk[n].bc.b; a.perm = w.bc_perm; a.out_off = fd::KW;
        if ((e = conv1d(L, "naive_bias_conv", a)) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t naive_convt(const Launch &L, int n, const float *x_in, float *x_out, int B, int Lin)
{
    const DevWeights &w = L.ctx->w;
    const int r = fd::ratio(n);
    const int64_t total = (int64_t)B * fd::C * Lin * r;
    FD_LAUNCH(L, "naive_convt", k_convt, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, x_in, w.blk[n].up.w, w.blk[n].up.b,
              x_out, B, Lin, r);
    return hipSuccess;
}

// one LVC layer, in place on x; y is scratch [B,32,Ln]
hipError_t naive_lvc_layer(const Launch &L, int n, int layer, float *x, const float *skip, float *y, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    const int hop = fd::hop(n), Ln = T * hop;
    const int64_t total = (int64_t)B * fd::C * Ln;
    const unsigned grid = (unsigned)((total + 255) / 256);
    FD_LAUNCH(L, "naive_add_skip", k_add_inplace, dim3(grid), dim3(256), 0, x, skip, total);
    ConvArgs a = {};
    a.B = B; a.Cin = fd::C; a.Cout = fd::C; a.L = Ln; a.in_stride = 1; a.in_len = Ln; a.KS = 3;
    int dil = 1;
    for (int i = 0; i < layer; ++i) dil *= 3;
    a.dil = dil; a.pre_slope = 0.2f; a.post_slope = 0.2f;
    a.x = x; a.w = w.blk[n].convs[layer].w; a.b = w.blk[n].convs[layer].b; a.out = y;
    hipError_t e = conv1d(L, "naive_lvc_conv", a);
    if (e != hipSuccess) return e;
    const float *kp = c->ws.kpack + (int64_t)n * B * T * fd::KREC;
    FD_LAUNCH(L, "naive_lvc_gate", k_lvc_gate, dim3(grid), dim3(256), 0, y, kp, x, B, T, hop, layer);
    return hipSuccess;
}

hipError_t naive_final_eps(const Launch &L, const float *x32, float *eps, int B, int T)
{
    const DevWeights &w = L.ctx->w;
    ConvArgs a = {};
    a.x = x32; a.w = w.final_.w; a.b = w.final_.b; a.out = eps;
    a.B = B; a.Cin = fd::C; a.Cout = 1; a.L = T * fd::HOPT; a.KS = 7; a.dil = 1; a.in_stride = 1; a.in_len = a.L;
    a.pre_slope = 1.0f; a.post_slope = 1.0f;
    return conv1d(L, "naive_final_conv", a);
}

}  // namespace fdk

// ------------------------------------------------------------------------------------------------
// kernels shared by both modes
// ------------------------------------------------------------------------------------------------
namespace fdk {

// a1 + a2: step embedding, 2-layer swish MLP, per-block fc_t  (util.py:407-432; FastDiff_model.py:85-87; modules.py:202)
// noise[s][b][blk][80].  A row is one distinct step value: the sampler's step s (the same for every utterance), or utterance b of
// fd_forward.  The MLP is 1.8 MB of weights against a few hundred outputs: one workgroup per row spent 57 us pulling them through
// one CU, so the second layer is spread over 8 workgroups per row (each redoes the small first layer) with the k range split
// over the waves, and the per-block layer runs as a second launch.
__global__ void __launch_bounds__(512) k_embed_mlp(const float *table, const float *w1T, const float *b1, const float *w2T,
                                                  const float *b2, const float *steps, const StepParams *params, int sampler,
                                                  float *h2g)
{
    __shared__ float emb[fd::E_IN], h1[fd::E_MID], part[8][64];
    const int slice = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    const float t = sampler ? params->table[r].t : steps[r];
    if (tid < 64) {
        const float arg = t * table[tid];
        emb[tid] = sinf(arg);
        emb[64 + tid] = cosf(arg);
    }
    __syncthreads();
    {
        float acc = b1[tid];
        for (int i = 0; i < fd::E_IN; ++i) acc += w1T[i * fd::E_MID + tid] * emb[i];
        h1[tid] = acc / (1.0f + expf(-acc));
    }
    __syncthreads();
    {
        const int o = slice * 64 + (tid & 63), kp = tid >> 6;
        float acc = 0.0f;
#pragma unroll 8
        for (int i = kp * 64; i < kp * 64 + 64; ++i) acc += w2T[i * fd::E_OUT + o] * h1[i];
        part[kp][tid & 63] = acc;
    }
    __syncthreads();
    if (tid < 64) {
        float acc = b2[slice * 64 + tid];
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) acc += part[kp][tid];
        h2g[r * fd::E_OUT + slice * 64 + tid] = acc / (1.0f + expf(-acc));
    }
}

// per-block fc_t on the rows of k_embed_mlp: thread = (k quarter, output); sampler rows are written for every utterance
__global__ void __launch_bounds__(1024) k_embed_fct(const float *h2g, const float *wt0, const float *bt0, const float *wt1,
                                                   const float *bt1, const float *wt2, const float *bt2, int sampler, float *noise, int B)
{
    __shared__ float h2[fd::E_OUT], part[4][256];
    const int r = blockIdx.x, tid = threadIdx.x, o240 = tid & 255, kp = tid >> 8;
    if (tid < fd::E_OUT) h2[tid] = h2g[r * fd::E_OUT + tid];
    __syncthreads();
    const int blk = o240 / fd::COND, o = o240 % fd::COND;
    if (o240 < fd::NBLK * fd::COND) {
        const float *wt = blk == 0 ? wt0 : (blk == 1 ? wt1 : wt2);
        float acc = 0.0f;
#pragma unroll 8
        for (int i = kp * 128; i < kp * 128 + 128; ++i) acc += wt[i * fd::COND + o] * h2[i];
        part[kp][o240] = acc;
    }
    __syncthreads();
    if (tid < fd::NBLK * fd::COND) {
        const float *bt = blk == 0 ? bt0 : (blk == 1 ? bt1 : bt2);
        const float v = bt[o] + part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        if (sampler) {
            for (int b = 0; b < B; ++b) noise[(((int64_t)r * B + b) * fd::NBLK + blk) * fd::COND + o] = v;
        } else {
            noise[((int64_t)r * fd::NBLK + blk) * fd::COND + o] = v;      // s = 0, b = r
        }
    }
}

hipError_t embed(const Launch &L, const StepIO &io, int B, int n_steps)
{
    const DevWeights &w = L.ctx->w;
    const int rows = io.sampler ? n_steps : B;
    float *h2g = L.ctx->ws.embed_h2;        // [max(1024, B) rows][512]
    FD_LAUNCH(L, "embed", k_embed_mlp, dim3(8, rows), dim3(512), 0, w.embed_table, w.fc_t1_T, w.fc_t1_b, w.fc_t2_T, w.fc_t2_b, io.steps,
              (const StepParams *)L.ctx->ws.params, io.sampler, h2g);
    FD_LAUNCH(L, "embed_fct", k_embed_fct, dim3(rows), dim3(1024), 0, (const float *)h2g, w.fc_t_T[0], w.fc_t_b[0], w.fc_t_T[1],
              w.fc_t_b[1], w.fc_t_T[2], w.fc_t_b[2], io.sampler, L.ctx->ws.noise, B);
    return hipSuccess;
}

// x [B][l4 float4s]; the draw of utterance b at offset off is keyed as in a batch of l4_io float4s per utterance (the caller's own
// length: the library's buffer may be padded to a frame bucket, fd_api.cpp) -- offsets behind l4_io are padding and stay untouched
__global__ void k_init_noise(float *x, int l4, int l4_io, unsigned long long seed, const unsigned long long *uids)
{
    const int off = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (off >= l4_io) return;
    const int64_t i = (int64_t)b * l4 + off;
    if (uids) reinterpret_cast<float4 *>(x)[i] = philox_normal4(seed, 0xFFFFFFFFu, (uint64_t)off, uids[b]);
    else reinterpret_cast<float4 *>(x)[i] = philox_normal4(seed, 0xFFFFFFFFu, (uint64_t)((int64_t)b * l4_io + off));
}

hipError_t init_noise(const Launch &L, float *x, int B, int l4, int l4_io, unsigned long long seed, const unsigned long long *uids)
{
    FD_LAUNCH(L, "init_noise", k_init_noise, dim3((unsigned)((l4_io + 255) / 256), B), dim3(256), 0, x, l4, l4_io, seed, uids);
    return hipSuccess;
}

// rows x width floats between two pitched buffers (pitches in floats): the library's frame-bucketed buffers <-> the caller's dense ones
// (blockIdx.z = replica: the same rows written `gridDim.z` times, rep_stride floats apart -- the hoisted predictor's batch holds the
// mel once per reverse step)
__global__ void k_copy_rows(float *__restrict__ dst, int64_t dpitch, const float *__restrict__ src, int64_t spitch, int width, int vec,
                            int64_t rep_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    dst += blockIdx.z * rep_stride;
    if (vec) {
        if (i * 4 < width) reinterpret_cast<float4 *>(dst + r * dpitch)[i] = reinterpret_cast<const float4 *>(src + r * spitch)[i];
    } else if (i < width) dst[r * dpitch + i] = src[r * spitch + i];
}

hipError_t copy_rows(const Launch &L, float *dst, int64_t dpitch, const float *src, int64_t spitch, int width, int rows, int reps, int64_t rep_stride)
{
    if (rows <= 0 || width <= 0 || reps <= 0) return hipSuccess;
    if (dpitch == width && spitch == width && reps == 1)
        return hipMemcpyAsync(dst, src, sizeof(float) * (size_t)width * rows, hipMemcpyDeviceToDevice, L.stream);
    const bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && ((dpitch | spitch | width | rep_stride) & 3) == 0;
    const int n = vec ? width / 4 : width;
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
        FD_LAUNCH(L, "copy_rows", k_copy_rows, dim3((unsigned)((n + 255) / 256), nr, reps), dim3(256), 0, dst + r0 * dpitch, dpitch, src + r0 * spitch, spitch,
                  width, vec ? 1 : 0, rep_stride);
    }
    return hipSuccess;
}

__global__ void k_update(float *x, const float *eps, const StepParams *p, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 xv = reinterpret_cast<const float4 *>(x)[i];
    const float4 ev = reinterpret_cast<const float4 *>(eps)[i];
    const int b = (int)(i / p->l4);          // (the naive path is never frame-bucketed: l4_io == l4)
    reinterpret_cast<float4 *>(x)[i] = sampler_update4(xv, ev, p, b, i - (int64_t)b * p->l4);
}

hipError_t naive_update(const Launch &L, float *x, const float *eps, int64_t n)
{
    const int64_t n4 = n / 4;
    FD_LAUNCH(L, "update", k_update, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, x, eps, L.ctx->ws.params, n4);
    return hipSuccess;
}

// End of a sampler step: next row of the step table; the fp16-range flags are cleared for the next step here rather than by a
// memset node (a captured hipMemsetAsync wrote garbage on the second replay of the graph on ROCm 7.2).
__global__ void k_advance(StepParams *p, int *range_flags, int inc, int set_step)
{
    if (threadIdx.x == 0) p->step_idx = inc ? p->step_idx + inc : set_step;
    if (threadIdx.x < 32) {      // this step's flags become "previous step" (inc == 0: start of a call or of a redo, all cleared)
        const int f = range_flags[threadIdx.x];
        range_flags[32 + threadIdx.x] = inc ? f : 0;
        range_flags[64 + threadIdx.x] = inc ? (range_flags[64 + threadIdx.x] | f) : 0;      // sticky over the call: what the host looks at
        range_flags[threadIdx.x] = 0;
    }
}

hipError_t advance_step(const Launch &L)
{
    FD_LAUNCH(L, "advance_step", k_advance, dim3(1), dim3(64), 0, L.ctx->ws.params, L.ctx->ws.range_flag, 1, 0);
    return hipSuccess;
}
hipError_t clear_range_flags(const Launch &L, int set_step)
{
    FD_LAUNCH(L, "clear_flags", k_advance, dim3(1), dim3(64), 0, L.ctx->ws.params, L.ctx->ws.range_flag, 0, set_step);
    return hipSuccess;
}

// ---- waveform epilogue: per-utterance abs-max, then /max * 32767 -> int16 (FastDiff.py:110; utils/audio.py:11-16)
__global__ void __launch_bounds__(256) k_absmax(const float *wav, int64_t len, unsigned int *maxbits, const long long *valid)
{
    __shared__ float wm[4];
    const int b = blockIdx.y;
    const int64_t n = valid ? (int64_t)valid[b] : len;          // ragged batch: only the utterance's own samples count
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(wav[(int64_t)b * len + i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup: thousands of them on B words serialise (measured 43 us for 7 MB with one per wave)
    if (threadIdx.x == 0) atomicMax(maxbits + b, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));   // non-negative floats order like uints
}

__global__ void k_to_int16(const float *wav, int64_t len, const unsigned int *maxbits, int16_t *pcm, const long long *valid)
{
    const int b = blockIdx.y;
    const float m = __uint_as_float(maxbits[b]);
    const int64_t n = valid ? (int64_t)valid[b] : len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = wav[(int64_t)b * len + i] / m;
        pcm[(int64_t)b * len + i] = i < n ? (int16_t)(v * 32767.0f) : (int16_t)0;      // silence behind a padded utterance
    }
}

hipError_t peak_normalize_int16(const Launch &L, const float *wav, int B, int64_t len, int16_t *pcm, const long long *valid_dev)
{
    unsigned int *maxbits = reinterpret_cast<unsigned int *>(L.ctx->scratch);   // [B] words
    hipError_t e = hipMemsetAsync(maxbits, 0, sizeof(unsigned int) * B, L.stream);
    if (e != hipSuccess) return e;
    const unsigned gx = (unsigned)((len + 256 * 8 - 1) / (256 * 8));
    FD_LAUNCH(L, "absmax", k_absmax, dim3((gx + 3) / 4, B), dim3(256), 0, wav, len, maxbits, valid_dev);
    FD_LAUNCH(L, "to_int16", k_to_int16, dim3(gx, B), dim3(256), 0, wav, len, (const unsigned int *)maxbits, pcm, valid_dev);
    return hipSuccess;
}

}  // namespace fdk
