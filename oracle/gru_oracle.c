/*
 * oracle/gru_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C float64 restatement of the OptiState GRU head `RNN.forward`
 * (reference gru/gru_model.py:25-49: nn.GRU(batch_first) with h0 = 0 -> last step ->
 * nn.Linear -> sigmoid).  The cell equations are torch.nn.GRU's published definition
 * (gate order r|z|n in the stacked weights):
 *     r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
 *     z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
 *     n = tanh  (W_in x + b_in + r * (W_hn h + b_hn))
 *     h' = (1 - z) * n + z * h
 * torch (third-party, pinned torch==1.13.1 in the reference's environment.yml:17, 2.10.0 in
 * this image) is a library, not reference source.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it against outputs of the
 * reference's own `RNN` class run on CPU in the build container (tools/gen_golden.py ->
 * tests/golden/gru_*.npz) to <= 2e-6 (torch computes in fp32, this file in fp64).
 *
 * Flat weight layout (also the C-ABI's, include/optistate_hip.h): for each layer l:
 *   W_ih_l [3H][I_l], W_hh_l [3H][H], b_ih_l [3H], b_hh_l [3H]; then fc.weight [C][H], fc.bias [C].
 * Also restated: the training target/loss of gru/gru_train.py:237-245 (ok_gru_train_loss).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double sigm(double v) { return 1.0 / (1.0 + exp(-v)); }

size_t ok_gru_param_count(int I, int H, int L, int C)
{
    size_t n = 0;
    for (int l = 0; l < L; l++) {
        int il = l == 0 ? I : H;
        n += (size_t)3 * H * il + (size_t)3 * H * H + 6 * (size_t)H;
    }
    return n + (size_t)C * H + C;
}

/* x [B][T][I]; out [B][C]; hlast (optional) [L][B][H]; seq_out (optional) [B][T][H] = top layer output. */
void ok_gru_forward(int B, int T, int I, int H, int L, int C, const double *x, const double *w,
                    int use_sigmoid, double *out, double *hlast, double *seq_out)
{
    /* trajectories are independent (one RNN.forward row each): optional OpenMP split over b, scratch per thread */
#pragma omp parallel
    {
    double *h = (double *)calloc((size_t)L * H, sizeof(double));
    double *hn = (double *)malloc(sizeof(double) * H);
    double *inp = (double *)malloc(sizeof(double) * (I > H ? I : H));
#pragma omp for schedule(static)
    for (int b = 0; b < B; b++) {
        memset(h, 0, sizeof(double) * L * H);
        for (int t = 0; t < T; t++) {
            int il = I;
            memcpy(inp, x + ((size_t)b * T + t) * I, sizeof(double) * I);
            const double *wp = w;
            for (int l = 0; l < L; l++) {
                const double *Wih = wp, *Whh = Wih + (size_t)3 * H * il, *bih = Whh + (size_t)3 * H * H, *bhh = bih + 3 * H;
                double *hl = h + (size_t)l * H;
                for (int j = 0; j < H; j++) {
                    double gi[3], gh[3];
                    for (int g = 0; g < 3; g++) {
                        double s = bih[g * H + j], u = bhh[g * H + j];
                        const double *wr = Wih + ((size_t)g * H + j) * il;
                        for (int k = 0; k < il; k++) s += wr[k] * inp[k];
                        const double *ur = Whh + ((size_t)g * H + j) * H;
                        for (int k = 0; k < H; k++) u += ur[k] * hl[k];
                        gi[g] = s; gh[g] = u;
                    }
                    double r = sigm(gi[0] + gh[0]);
                    double z = sigm(gi[1] + gh[1]);
                    double n = tanh(gi[2] + r * gh[2]);
                    hn[j] = (1.0 - z) * n + z * hl[j];
                }
                memcpy(hl, hn, sizeof(double) * H);
                memcpy(inp, hn, sizeof(double) * H);
                wp = bhh + 3 * H;
                il = H;
            }
            if (seq_out) memcpy(seq_out + ((size_t)b * T + t) * H, h + (size_t)(L - 1) * H, sizeof(double) * H);
        }
        const double *fw = w + ok_gru_param_count(I, H, L, C) - ((size_t)C * H + C), *fb = fw + (size_t)C * H;
        const double *top = h + (size_t)(L - 1) * H;
        for (int c = 0; c < C; c++) {
            double s = fb[c];
            for (int k = 0; k < H; k++) s += fw[(size_t)c * H + k] * top[k];
            out[(size_t)b * C + c] = use_sigmoid ? sigm(s) : s;
        }
        if (hlast)
            for (int l = 0; l < L; l++) memcpy(hlast + ((size_t)l * B + b) * H, h + (size_t)l * H, sizeof(double) * H);
    }
    free(h); free(hn); free(inp);
    }
}

/* gru/gru_train.py:237-245: target = [y(12), |out[0:12] - y|(12)] with out detached; loss = mean
 * squared error over all B*24 entries.  out [B][24], y [B][12]; target_out (optional) [B][24]. */
double ok_gru_train_loss(int B, const double *out, const double *y, double *target_out)
{
    double acc = 0.0;
    for (int b = 0; b < B; b++)
        for (int c = 0; c < 24; c++) {
            double tgt = c < 12 ? y[b * 12 + c] : fabs(out[b * 24 + (c - 12)] - y[b * 12 + (c - 12)]);
            double d = out[b * 24 + c] - tgt;
            acc += d * d;
            if (target_out) target_out[b * 24 + c] = tgt;
        }
    return acc / ((double)B * 24.0);
}

/* Feature pack + min-max normalise (data_collection/data_conversion_Kalman_to_Training.py:245-254,
 * gru/gru_test.py:99-101): row = [x_post 12 | accel 6 | f 12 | p_world 12 | dp 12 | imu 6];
 * (row - min) / (max - min). */
void ok_feature_row(const double *x_post, const double *accel, const double *f, const double *p_world,
                    const double *dp, const double *imu, const double *minv, const double *maxv, double *row)
{
    int o = 0;
    for (int i = 0; i < 12; i++) row[o++] = x_post[i];
    for (int i = 0; i < 6; i++) row[o++] = accel[i];
    for (int i = 0; i < 12; i++) row[o++] = f[i];
    for (int i = 0; i < 12; i++) row[o++] = p_world[i];
    for (int i = 0; i < 12; i++) row[o++] = dp[i];
    for (int i = 0; i < 6; i++) row[o++] = imu[i];
    if (minv && maxv)
        for (int i = 0; i < 60; i++) row[i] = (row[i] - minv[i]) / (maxv[i] - minv[i]);
}
