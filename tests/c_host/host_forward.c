/* host_forward.c -- the C ABI of include/desire_hip.h driven from plain C (no Python, no torch): reads a blob of dims,
 * weights and inputs, runs desire_forward on device buffers it owns, writes the refined trajectories and scores.
 * Built by tests/test_c_host.py with gcc (the header is plain C) against libdesire_hip.so and the HIP runtime.
 *
 * blob:  desire_dims | int32 n_weights | n_weights x { int32 name_len, name, int64 n, float[n] }
 *        | past [n_scenes,T_obs,mno,3] | fut [n_scenes,T_pred,mno,3] | eps [R,L] | grids [n_grids,Gh,Gw,C] | int32 gos[n_scenes]
 * out:   Y [R,T_pred,2] | score [R] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "desire_hip.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "host_forward: %s failed (%s)\n", #x, desire_last_error()); return 2; } } while (0)

static float* to_device(const float* host, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: host_forward in.blob out.bin\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    CHECK(f != NULL);
    desire_dims d;
    CHECK(desire_dims_size() == (int)sizeof(desire_dims));     /* this host and the library agree about the struct */
    CHECK(fread(&d, sizeof(d), 1, f) == 1);
    desire_handle* h = NULL;
    CHECK(desire_create(&d, &h) == DESIRE_OK);
    int32_t nw = 0;
    CHECK(fread(&nw, 4, 1, f) == 1);
    for (int i = 0; i < nw; ++i) {
        int32_t len; char name[256]; int64_t n;
        CHECK(fread(&len, 4, 1, f) == 1 && len < 256);
        CHECK(fread(name, 1, (size_t)len, f) == (size_t)len);
        name[len] = 0;
        CHECK(fread(&n, 8, 1, f) == 1);
        float* w = (float*)malloc((size_t)n * sizeof(float));
        CHECK(w != NULL && fread(w, sizeof(float), (size_t)n, f) == (size_t)n);
        CHECK(desire_set_weight(h, name, w, (size_t)n) == DESIRE_OK);
        free(w);
    }
    CHECK(desire_finalize_weights(h) == DESIRE_OK);
    const size_t R = (size_t)d.n_scenes * d.K * d.mno;
    const size_t n_past = (size_t)d.n_scenes * d.T_obs * d.mno * 3, n_fut = (size_t)d.n_scenes * d.T_pred * d.mno * 3;
    const size_t n_eps = R * d.L, n_grids = (size_t)d.n_grids * d.Gh * d.Gw * d.C;
    size_t n_all = n_past + n_fut + n_eps + n_grids;
    float* host = (float*)malloc(n_all * sizeof(float));
    CHECK(host != NULL && fread(host, sizeof(float), n_all, f) == n_all);
    int32_t* gos = (int32_t*)malloc((size_t)d.n_scenes * 4);
    CHECK(gos != NULL && fread(gos, 4, (size_t)d.n_scenes, f) == (size_t)d.n_scenes);
    fclose(f);
    float* dev_past = to_device(host, n_past);
    float* dev_fut = to_device(host + n_past, n_fut);
    float* dev_eps = to_device(host + n_past + n_fut, n_eps);
    float* dev_grids = to_device(host + n_past + n_fut + n_eps, n_grids);
    float *dev_Y = NULL, *dev_score = NULL;
    CHECK(dev_past && dev_fut && dev_eps && dev_grids);
    CHECK(hipMalloc((void**)&dev_Y, R * d.T_pred * 2 * sizeof(float)) == hipSuccess);
    CHECK(hipMalloc((void**)&dev_score, R * sizeof(float)) == hipSuccess);
    hipStream_t s;
    CHECK(hipStreamCreate(&s) == hipSuccess);
    CHECK(desire_set_scene_grids(h, dev_grids, gos) == DESIRE_OK);
    CHECK(desire_forward(h, dev_past, dev_fut, dev_eps, dev_Y, dev_score, (void*)s) == DESIRE_OK);
    CHECK(hipStreamSynchronize(s) == hipSuccess);
    float* out = (float*)malloc((R * d.T_pred * 2 + R) * sizeof(float));
    CHECK(out != NULL);
    CHECK(hipMemcpy(out, dev_Y, R * d.T_pred * 2 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(out + R * d.T_pred * 2, dev_score, R * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    FILE* g = fopen(argv[2], "wb");
    CHECK(g != NULL && fwrite(out, sizeof(float), R * d.T_pred * 2 + R, g) == R * d.T_pred * 2 + R);
    fclose(g);
    /* error behaviour from C: a bad weight name is refused with a message, the handle stays usable */
    CHECK(desire_set_weight(h, "no/such/weight", host, 1) == DESIRE_ERR_ARG && strlen(desire_last_error()) > 0);
    CHECK(desire_destroy(h) == DESIRE_OK);
    printf("host_forward: %zu rows ok\n", R);
    return 0;
}
