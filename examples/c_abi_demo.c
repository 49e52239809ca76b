/* Plain-C host of liboptistate_hip.so: the drop-in boundary used WITHOUT Python or torch -- hipMalloc'd buffers, the C-ABI calls of
 * include/optistate_hip.h, results copied back.  It runs the three calls a host of the reference's pipeline would make:
 *   os_kf_step   one filter step of one Kalman_Filter instance on host float64 arrays (the caller loop of
 *                data_collection/data_conversion_Kalman_to_Training.py:193-199),
 *   os_kf_run    B trajectories x T steps of get_odom / set_measurements / predict / update (kalman_filter/kalman_filter.py:79-174),
 *   os_fused_run the same plus feature rows, min-max normalisation and RNN.forward (gru/gru_model.py:25-49) in one call.
 * Input and output are raw little-endian files so that a test can feed it seeded data and compare with the float64 oracle
 * (tests/test_gpu_c_host.py).
 *
 *   build: gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_demo.c \
 *              -Loptistate_amd/lib -loptistate_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/optistate_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   run:   ./c_abi_demo in.bin out.bin
 *
 * in.bin : int32 B, T, L (GRU(60, 64, L, 24)), n_w; float p[T][12][B], f, dp; imu[T][6][B]; uint32 contact[T][B]; accel[T][6][B];
 *          x0[12][B]; P0[144][B]; Q[144]; R[100]; minmax[2][60]; w_flat[n_w]
 * out.bin: float x_out[T][12][B] (os_kf_run); int32 status[B]; float x_out[T][12][B] (os_fused_run); float out[B][24]; int32 status[B];
 *          double x[12], z[10] (os_kf_step of trajectory 0, step 0, in float64)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "optistate_hip.h"

#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_OS(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, os_last_error(ctx)); return 3; } } while (0)

static void *read_block(FILE *fp, size_t bytes)
{
    void *h = malloc(bytes);
    if (!h || fread(h, 1, bytes, fp) != bytes) { fprintf(stderr, "short input file\n"); exit(4); }
    return h;
}

/* host block -> device copy (the library takes device pointers) */
static int to_device(const void *h, size_t bytes, void **d)
{
    CHECK_HIP(hipMalloc(d, bytes));
    CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 1; }
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, fp) != 4) { fprintf(stderr, "short header\n"); return 4; }
    const int32_t B = hdr[0], T = hdr[1], L = hdr[2], n_w = hdr[3];
    const size_t s12 = (size_t)T * 12 * B * sizeof(float), s6 = (size_t)T * 6 * B * sizeof(float);
    float *p = read_block(fp, s12), *f = read_block(fp, s12), *dp = read_block(fp, s12), *imu = read_block(fp, s6);
    uint32_t *contact = read_block(fp, (size_t)T * B * sizeof(uint32_t));
    float *accel = read_block(fp, s6);
    float *x0 = read_block(fp, (size_t)12 * B * sizeof(float)), *P0 = read_block(fp, (size_t)144 * B * sizeof(float));
    float *Q = read_block(fp, 144 * sizeof(float)), *R = read_block(fp, 100 * sizeof(float));
    float *minmax = read_block(fp, 120 * sizeof(float)), *w = read_block(fp, (size_t)n_w * sizeof(float));
    fclose(fp);

    /* settings.py:5,11,20-23 (the inertia is given there in g mm^2: 55303643.08, 60119440.34, 105304340.05, divided by 1e9) */
    os_kf_config cfg = {0, 0.01f, 8.8f, {55303643.08f / 1e9f, 60119440.34f / 1e9f, 105304340.05f / 1e9f}, -9.81f};
    os_ctx *ctx = NULL;
    if (os_create(&cfg, &ctx) != 0 || !ctx) { fprintf(stderr, "os_create failed\n"); return 3; }
    printf("liboptistate_hip version %d, built for %s, build id %s\n", os_version(), os_build_arch(), os_build_id());
    CHECK_OS(os_kf_set_noise(ctx, Q, R));

    void *d_p, *d_f, *d_dp, *d_imu, *d_c, *d_acc, *d_x, *d_P, *d_mm, *d_w, *d_xout, *d_st, *d_out;
    if (to_device(p, s12, &d_p) || to_device(f, s12, &d_f) || to_device(dp, s12, &d_dp) || to_device(imu, s6, &d_imu) ||
        to_device(contact, (size_t)T * B * 4, &d_c) || to_device(accel, s6, &d_acc) || to_device(minmax, 480, &d_mm) ||
        to_device(w, (size_t)n_w * 4, &d_w) || to_device(x0, (size_t)48 * B, &d_x) || to_device(P0, (size_t)576 * B, &d_P)) return 2;
    CHECK_HIP(hipMalloc(&d_xout, s12));
    CHECK_HIP(hipMalloc(&d_st, (size_t)B * 4));
    CHECK_HIP(hipMalloc(&d_out, (size_t)B * 24 * 4));

    fp = fopen(argv[2], "wb");
    if (!fp) { perror(argv[2]); return 1; }
    float *h_xout = malloc(s12), *h_out = malloc((size_t)B * 24 * 4);
    int32_t *h_st = malloc((size_t)B * 4);

    /* ---- B x T filter steps ---- */
    CHECK_OS(os_kf_run(ctx, B, T, d_p, d_f, d_dp, d_imu, d_c, NULL, d_x, d_P, d_xout, NULL, NULL, NULL, d_st,
                       OS_KF_SEQUENTIAL_UPDATE | OS_KF_SYMMETRIC_P, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h_xout, d_xout, s12, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_st, d_st, (size_t)B * 4, hipMemcpyDeviceToHost));
    fwrite(h_xout, 1, s12, fp);
    fwrite(h_st, 4, (size_t)B, fp);

    /* ---- the fused call: filter + feature rows + normalisation + GRU head (x, P start again from x0, P0) ---- */
    os_gru_dims gd = {60, 64, L, 24, 1};
    if (os_gru_param_count(&gd) != (size_t)n_w) { fprintf(stderr, "weight count %d != %zu\n", n_w, os_gru_param_count(&gd)); return 4; }
    CHECK_OS(os_gru_load(ctx, &gd, d_w, NULL));
    CHECK_HIP(hipMemcpy(d_x, x0, (size_t)48 * B, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_P, P0, (size_t)576 * B, hipMemcpyHostToDevice));
    CHECK_OS(os_fused_run(ctx, B, T, d_p, d_f, d_dp, d_imu, d_c, d_acc, NULL, NULL, 0, d_mm, d_x, d_P, d_xout, d_out, d_st,
                          OS_KF_SEQUENTIAL_UPDATE | OS_KF_SYMMETRIC_P, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h_xout, d_xout, s12, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_out, d_out, (size_t)B * 24 * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_st, d_st, (size_t)B * 4, hipMemcpyDeviceToHost));
    fwrite(h_xout, 1, s12, fp);
    fwrite(h_out, 4, (size_t)B * 24, fp);
    fwrite(h_st, 4, (size_t)B, fp);

    /* ---- one step of one filter instance on host float64 arrays (trajectory 0, step 0) ---- */
    double sp[12], sf[12], sdp[12], simu[6], sx[12], sP[144], sQ[144], sR[100], sz[10];
    uint8_t sc[4];
    int32_t sst = 0;
    for (int i = 0; i < 12; i++) { sp[i] = p[(size_t)i * B]; sf[i] = f[(size_t)i * B]; sdp[i] = dp[(size_t)i * B]; sx[i] = x0[(size_t)i * B]; }
    for (int i = 0; i < 6; i++) simu[i] = imu[(size_t)i * B];
    for (int i = 0; i < 144; i++) { sP[i] = P0[(size_t)i * B]; sQ[i] = Q[i]; }
    for (int i = 0; i < 100; i++) sR[i] = R[i];
    for (int k = 0; k < 4; k++) sc[k] = (uint8_t)((contact[0] >> (8 * k)) & 0xff);
    const double model[6] = {0.01, 8.8, 55303643.08 / 1e9, 60119440.34 / 1e9, 105304340.05 / 1e9, -9.81};      /* float64, like the reference's */
    CHECK_OS(os_kf_step(ctx, OS_STEP_ODOM | OS_STEP_PREDICT | OS_STEP_UPDATE, model, sp, sf, sdp, simu, sc, NULL, sQ, sR, sx, sP, sz,
                        NULL, NULL, NULL, NULL, NULL, &sst, NULL));
    fwrite(sx, sizeof(double), 12, fp);
    fwrite(sz, sizeof(double), 10, fp);
    fclose(fp);

    int bad = 0;
    for (int b = 0; b < B; b++) bad += (h_st[b] & OS_STATUS_FAIL_MASK) != 0;
    printf("B = %d, T = %d: os_kf_run, os_fused_run (GRU(60,64,%d,24)) and os_kf_step done; failed trajectories %d, step status %d\n", B, T, L, bad, sst);
    os_destroy(ctx);
    return bad || (sst & OS_STATUS_FAIL_MASK) ? 5 : 0;
}
