/*
 * c_host.c -- a host program in plain C that drives libfastdiff_hip.so through its C ABI alone (no Python, no torch):
 * what a binding in any language with a C FFI amounts to.  See INTEGRATION.md, section B.
 *
 *   cc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host.c -o examples/c_host \
 *      -Lfastdiff_amd/lib -lfastdiff_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/fastdiff_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   examples/c_host job.bin out.f32 [pad_MiB [key=value,...]]
 *   (pad_MiB, optional: reserve that much device memory before anything else -- a testing aid that moves the addresses of everything
 *   the library allocates afterwards; tools/xproc_hunt.py uses it.  key=value,...: library options, fd_set_option)
 *
 * job.bin (little endian, written by tests/test_c_host.py):
 *   int32 n_tensors; per tensor: int32 name_len, name bytes, int32 ndim, int64 dims[ndim], float data[]      -- the state_dict
 *   int32 B, T, N, ddim;  fd_step table[N];  float mel[B*80*T];  float x_T[B*T*256];  float z[N*B*T*256]
 * out.f32: float x_0[B*T*256] = sampling_given_noise_schedule(...) with the injected noise (util.py:158-235).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include "fastdiff_hip.h"
#include "fastdiff_hip_ext.h"      /* fd_get_counter: only to SHOW what the range check did; the sampling itself needs fastdiff_hip.h alone */

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define FD(h, x) do { int rc_ = (x); if (rc_ != FD_OK) DIE("%s -> %d: %s", #x, rc_, fd_last_error(h)); } while (0)

static void rd(FILE *f, void *dst, size_t n) { if (fread(dst, 1, n, f) != n) DIE("job file too short"); }

int main(int argc, char **argv)
{
    if (argc < 3 || argc > 5) DIE("usage: %s job.bin out.f32 [pad_MiB [key=value,...]]", argv[0]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    if (argc >= 4 && atol(argv[3]) > 0) {
        void *pad;
        HIP(hipMalloc(&pad, (size_t)atol(argv[3]) << 20));      /* kept until exit */
    }

    fd_config cfg;
    fd_handle h = NULL;
    fd_default_config(&cfg);
    FD(NULL, fd_create(&cfg, 0, &h));
    if (fd_abi_revision() < 2) DIE("this host reads fd_sample's result without fd_sample_check: it needs ABI revision >= 2");
    if (argc == 5) {
        char *opts = (char *)malloc(strlen(argv[4]) + 1);      /* (strdup is not C99) */
        strcpy(opts, argv[4]);
        for (char *kv = strtok(opts, ","); kv; kv = strtok(NULL, ",")) {
            char *eq = strchr(kv, '=');
            if (!eq) DIE("option '%s' is not key=value", kv);
            *eq = 0;
            FD(h, fd_set_option(h, kv, eq + 1));
        }
        free(opts);
    }

    int32_t n_tensors;
    rd(f, &n_tensors, 4);
    for (int i = 0; i < n_tensors; ++i) {
        int32_t name_len, ndim;
        char name[256];
        int64_t dims[8], n = 1;
        rd(f, &name_len, 4);
        if (name_len <= 0 || name_len >= (int)sizeof(name)) DIE("bad name length");
        rd(f, name, (size_t)name_len);
        name[name_len] = 0;
        rd(f, &ndim, 4);
        if (ndim < 1 || ndim > 8) DIE("bad rank for %s", name);
        rd(f, dims, sizeof(int64_t) * (size_t)ndim);
        for (int d = 0; d < ndim; ++d) n *= dims[d];
        float *w = (float *)malloc(sizeof(float) * (size_t)n);
        rd(f, w, sizeof(float) * (size_t)n);
        FD(h, fd_set_weight(h, name, w, dims, ndim));          /* copied before return */
        free(w);
    }
    FD(h, fd_commit_weights(h));                                /* folds weight-norm, packs for the matrix cores, uploads */

    int32_t hdr[4];
    rd(f, hdr, sizeof(hdr));
    const int B = hdr[0], T = hdr[1], N = hdr[2], ddim = hdr[3];
    const size_t L = (size_t)T * 256, n_mel = (size_t)B * 80 * T, n_x = (size_t)B * L;
    fd_step *table = (fd_step *)malloc(sizeof(fd_step) * (size_t)N);
    float *host = (float *)malloc(sizeof(float) * n_x * (size_t)(N > 1 ? N : 1));
    float *mel_d, *xT_d, *z_d, *out_d;
    rd(f, table, sizeof(fd_step) * (size_t)N);
    HIP(hipMalloc((void **)&mel_d, sizeof(float) * n_mel));
    HIP(hipMalloc((void **)&xT_d, sizeof(float) * n_x));
    HIP(hipMalloc((void **)&z_d, sizeof(float) * n_x * (size_t)N));
    HIP(hipMalloc((void **)&out_d, sizeof(float) * n_x));
    rd(f, host, sizeof(float) * n_mel);
    HIP(hipMemcpy(mel_d, host, sizeof(float) * n_mel, hipMemcpyHostToDevice));
    rd(f, host, sizeof(float) * n_x);
    HIP(hipMemcpy(xT_d, host, sizeof(float) * n_x, hipMemcpyHostToDevice));
    rd(f, host, sizeof(float) * n_x * (size_t)N);
    HIP(hipMemcpy(z_d, host, sizeof(float) * n_x * (size_t)N, hipMemcpyHostToDevice));
    fclose(f);

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    /* call, synchronise, read -- the reference's contract (util.py:215-235).  With the library's defaults fd_sample has looked at its own
     * fp16-range check by the time it returns (and has run the call again on the fp32 kernels if an operand did not fit): nothing else
     * to call before out_d is read.  (The pipelined form -- option defer_check = 1, fd_sample_settle -- is opt-in.) */
    FD(h, fd_sample(h, mel_d, B, T, NULL, table, N, ddim, xT_d, z_d, 0, out_d, NULL, stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(host, out_d, sizeof(float) * n_x, hipMemcpyDeviceToHost));

    FILE *o = fopen(argv[2], "wb");
    if (!o || fwrite(host, sizeof(float), n_x, o) != n_x) DIE("cannot write %s", argv[2]);
    fclose(o);
    double acc = 0.0;
    for (size_t i = 0; i < n_x; ++i) acc += host[i] >= 0 ? host[i] : -host[i];
    printf("%s: B=%d T=%d N=%d  mean|x_0| = %.6f  calls_redone=%lld\n", fd_version(), B, T, N, acc / (double)n_x,
           (long long)fd_get_counter(h, "calls_redone"));
    FD(h, fd_destroy(h));
    return 0;
}
