// vit_kernels.hip -- encoder half of the reference's ViT auto-encoder (transformer/transformer_model.py:113-135; optional
// row A10 / BASELINE config 5): depth frames (N,1,224,224) -> 128-d latent in (0,1) that is appended to the 60 Kalman
// features (gru/gru_test.py:119-136).  The blocks are timm-0.3.2 `Block`s (pre-LN, fused qkv with bias, GELU(erf) MLP,
// LayerNorm eps 1e-5, no drop-path); timm itself is absent from the build image, so this follows the published
// definition of that release -- parity UNPINNED, checked only against this repo's own float64
// restatement (oracle/vit_oracle.py).
//
// Dense projections ([N*197 x 128] x [128 x 384/128/512], [.. x 512] x [512 x 128], patch embedding [N*196 x 256] x
// [256 x 128]) are plain library GEMMs -> rocBLAS sgemm.  Hand-written here: patch extraction, token assembly (+cls,
// +pos), LayerNorm, per-(image, head) attention with K/V in LDS and an online softmax per query lane, bias+GELU,
// bias+residual, final LayerNorm+sigmoid of the cls token.
#include "launch.hpp"

#include <math.h>
#include <rocblas/rocblas.h>

namespace osv {

struct VitDims { int img, patch, dim, depth, heads, mlp; };

__host__ __device__ inline int grid_of(const VitDims &d) { return d.img / d.patch; }
__host__ __device__ inline int ntok(const VitDims &d) { return grid_of(d) * grid_of(d) + 1; }

// images [N][img][img] -> patches [N*G*G][patch*patch] (row = one patch, pixel order (py, px) as Conv2d flattens it)
__global__ void patchify_kernel(int N, VitDims d, const float *img, float *out)
{
    const int G = grid_of(d), PP = d.patch * d.patch;
    const size_t total = (size_t)N * G * G * PP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pix = i % PP;
        const size_t pr = i / PP;
        const int gx = pr % G, gy = (pr / G) % G;
        const size_t n = pr / ((size_t)G * G);
        const int py = pix / d.patch, px = pix % d.patch;
        out[i] = img[(n * d.img + (size_t)gy * d.patch + py) * d.img + (size_t)gx * d.patch + px];
    }
}

// X [N][L][D]: row 0 = cls + pos[0]; row 1+i = tok[n][i] + patch_b + pos[1+i]   (transformer_model.py:115-123)
__global__ void assemble_kernel(int N, int L, int D, const float *tok, const float *patch_b, const float *cls, const float *pos,
                                float *X)
{
    const size_t total = (size_t)N * L * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % D, l = (i / D) % L;
        const size_t n = i / ((size_t)L * D);
        X[i] = (l == 0 ? cls[c] : tok[(n * (L - 1) + (l - 1)) * D + c] + patch_b[c]) + pos[(size_t)l * D + c];
    }
}

// LayerNorm over rows of D (D <= 256, one wave per row, eps 1e-5)
__global__ void layernorm_kernel(size_t M, int D, const float *X, const float *w, const float *b, float *Y, size_t x_stride)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *x = X + row * x_stride;
    float v[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; v[j] = c < D ? x[c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; const float t = c < D ? v[j] - mean : 0.f; q += t * t; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; if (c < D) Y[row * D + c] = (v[j] - mean) * rstd * w[c] + b[c]; }
}

// Attention for one (image, head): QKV rows [L][3D] (+bias), head h owns columns h*hd..; K and V of the head staged in LDS,
// one query per lane with an online softmax; output O[row][h*hd + :].  hd <= 64.
template <int HD>
__global__ void attention_kernel(int L, int D, int heads, const float *QKV, const float *qkv_b, float *O)
{
    extern __shared__ float sm[];            // K [L][HD+1], V [L][HD+1]
    float *Ks = sm, *Vs = sm + (size_t)L * (HD + 1);
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *base = QKV + (size_t)n * L * 3 * D;
    for (int i = threadIdx.x; i < L * HD; i += blockDim.x) {
        const int l = i / HD, c = i % HD;
        Ks[l * (HD + 1) + c] = base[(size_t)l * 3 * D + D + h * HD + c] + qkv_b[D + h * HD + c];
        Vs[l * (HD + 1) + c] = base[(size_t)l * 3 * D + 2 * D + h * HD + c] + qkv_b[2 * D + h * HD + c];
    }
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    for (int qi = threadIdx.x; qi < L; qi += blockDim.x) {
        float q[HD], acc[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) { q[c] = (base[(size_t)qi * 3 * D + h * HD + c] + qkv_b[h * HD + c]) * scale; acc[c] = 0.f; }
        float mx = -3.0e38f, den = 0.f;
        for (int l = 0; l < L; l++) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; c++) s += q[c] * Ks[l * (HD + 1) + c];
            const float mn = fmaxf(mx, s);
            const float corr = __expf(mx - mn), p = __expf(s - mn);
            den = den * corr + p;
#pragma unroll
            for (int c = 0; c < HD; c++) acc[c] = acc[c] * corr + p * Vs[l * (HD + 1) + c];
            mx = mn;
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int c = 0; c < HD; c++) O[((size_t)n * L + qi) * D + h * HD + c] = acc[c] * inv;
    }
}

// The same attention on the matrix cores for head_dim = 32 (the reference's ViT: 4 heads x 32).  One workgroup per
// (image, head), K and V (+bias) of the head in LDS; a wave owns 32-query tiles.  Per tile: S = Q K^T as seven 32x32
// v_mfma_f32_32x32x2_f32 tiles (exact fp32), keys beyond L masked, row max / row sum by in-lane reduction over the tiles
// plus five xor-shuffles inside each 32-lane half (a C-layout row lives in one half), P = exp(S - max) handed from the
// C layout to the A layout through a wave-private LDS tile, O = P V accumulated over the key tiles, scaled by 1 / sum.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int ATT_HD = 32, ATT_KS = ATT_HD + 1, ATT_MAXT = 8;      // up to 256 tokens

template <int NT /* key / query tiles = ceil(L / 32) */>
__global__ __launch_bounds__(256, 1) void attention_mfma_kernel(int L, int D, int heads, const float *QKV, const float *qkv_b, float *O)
{
    extern __shared__ float sm[];            // K [L][33] | V [L][33] | P tiles [4][32][33]
    float *Ks = sm, *Vs = sm + (size_t)L * ATT_KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    float *pt = sm + (size_t)2 * L * ATT_KS + (size_t)wave * 32 * ATT_KS;
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *base = QKV + (size_t)n * L * 3 * D;
    // staging with float4 loads, four in flight per thread (a scalar loop here was one exposed HBM round trip per element:
    // it dominated the kernel)
    {
        const int c4 = (threadIdx.x & 7) * 4;                       // 8 float4 per 32-wide head row
        const float4 kb = *reinterpret_cast<const float4 *>(qkv_b + D + h * ATT_HD + c4);
        const float4 vb = *reinterpret_cast<const float4 *>(qkv_b + 2 * D + h * ATT_HD + c4);
#pragma unroll 4
        for (int l = threadIdx.x >> 3; l < L; l += 32) {
            const float4 kv = *reinterpret_cast<const float4 *>(base + (size_t)l * 3 * D + D + h * ATT_HD + c4);
            const float4 vv = *reinterpret_cast<const float4 *>(base + (size_t)l * 3 * D + 2 * D + h * ATT_HD + c4);
            float *kd = Ks + l * ATT_KS + c4, *vd = Vs + l * ATT_KS + c4;
            kd[0] = kv.x + kb.x; kd[1] = kv.y + kb.y; kd[2] = kv.z + kb.z; kd[3] = kv.w + kb.w;
            vd[0] = vv.x + vb.x; vd[1] = vv.y + vb.y; vd[2] = vv.z + vb.z; vd[3] = vv.w + vb.w;
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)ATT_HD);
    for (int qt = wave; qt < NT; qt += 4) {
        const int r0 = 32 * qt;
        // Q tile as A fragments: lane -> (row r0 + li, dim 2q + lh)
        const int qrow = r0 + li < L ? r0 + li : L - 1;
        float qa[ATT_HD / 2];
#pragma unroll
        for (int q = 0; q < ATT_HD / 2; q++)
            qa[q] = (base[(size_t)qrow * 3 * D + h * ATT_HD + 2 * q + lh] + qkv_b[h * ATT_HD + 2 * q + lh]) * scale;
        // S tiles; the 16 K fragments of tile jt+1 are requested from LDS before the 16 MFMAs of tile jt issue
        f32x16v S[NT];
        float kb[2][ATT_HD / 2];
        {
            const int key = li < L ? li : L - 1;
#pragma unroll
            for (int q = 0; q < ATT_HD / 2; q++) kb[0][q] = Ks[key * ATT_KS + 2 * q + lh];
        }
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            if (jt + 1 < NT) {
                const int key = 32 * (jt + 1) + li < L ? 32 * (jt + 1) + li : L - 1;
#pragma unroll
                for (int q = 0; q < ATT_HD / 2; q++) kb[(jt + 1) & 1][q] = Ks[key * ATT_KS + 2 * q + lh];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) S[jt][e] = 0.f;
#pragma unroll
            for (int q = 0; q < ATT_HD / 2; q++) S[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q], kb[jt & 1][q], S[jt], 0, 0, 0);
            if (32 * jt + li >= L) {                     // this lane's key column does not exist
#pragma unroll
                for (int e = 0; e < 16; e++) S[jt][e] = -3.0e38f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // row statistics: element e of a lane belongs to row (e&3) + 8(e>>2) + 4 lh, its 32 columns sit in the 32 lanes of the half
        float mx[16], den[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            float m = S[0][e];
#pragma unroll
            for (int jt = 1; jt < NT; jt++) m = fmaxf(m, S[jt][e]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            mx[e] = m;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            float d = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; jt++) { const float pv = __expf(S[jt][e] - mx[e]); S[jt][e] = pv; d += pv; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            den[e] = d;
        }
        // O = P V
        f32x16v acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        float vbf[2][16];
        {
#pragma unroll
            for (int q = 0; q < 16; q++) { const int key = 2 * q + lh < L ? 2 * q + lh : L - 1; vbf[0][q] = Vs[key * ATT_KS + li]; }
        }
#pragma unroll
        for (int jt = 0; jt < NT; jt++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++) pt[((e & 3) + 8 * (e >> 2) + 4 * lh) * ATT_KS + li] = S[jt][e];
            __builtin_amdgcn_wave_barrier();
            float pa[16];
#pragma unroll
            for (int q = 0; q < 16; q++) pa[q] = pt[li * ATT_KS + 2 * q + lh];
            if (jt + 1 < NT) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int key = 32 * (jt + 1) + 2 * q + lh < L ? 32 * (jt + 1) + 2 * q + lh : L - 1;      // P is 0 there
                    vbf[(jt + 1) & 1][q] = Vs[key * ATT_KS + li];
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[q], vbf[jt & 1][q], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row < L) O[((size_t)n * L + row) * D + h * ATT_HD + li] = acc[e] / den[e];
        }
    }
}

// X += P + bias
__global__ void bias_residual_kernel(size_t total, int D, const float *P, const float *bias, float *X)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        X[i] += P[i] + bias[i % D];
}

// H = gelu(H + bias), exact erf form (nn.GELU default of that release)
__global__ void bias_gelu_kernel(size_t total, int D, const float *bias, float *Hm)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = Hm[i] + bias[i % D];
        Hm[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    }
}

// latent[n] = sigmoid(LN(X[n][0]))   (transformer_model.py:127-133)
__global__ void cls_head_kernel(int N, int L, int D, const float *X, const float *w, const float *b, float *latent)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *x = X + (size_t)n * L * D;
    float v[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; v[j] = c < D ? x[c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = lane + 64 * j; const float t = c < D ? v[j] - mean : 0.f; q += t * t; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = lane + 64 * j;
        if (c < D) latent[(size_t)n * D + c] = 1.0f / (1.0f + expf(-((v[j] - mean) * rstd * w[c] + b[c])));
    }
}

}  // namespace osv

using namespace osv;

struct os_vit_state {
    VitDims d;
    const float *w;          // caller-owned flat weights
    rocblas_handle blas;
    float *buf; size_t buf_floats;
    bool att_attr_set;
};

static size_t vit_param_count(const VitDims &d)
{
    const size_t D = d.dim, PP = (size_t)d.patch * d.patch, L = ntok(d), Mh = d.mlp;
    size_t n = D * PP + D + D + L * D;
    n += (size_t)d.depth * (2 * D + 3 * D * D + 3 * D + D * D + D + 2 * D + Mh * D + Mh + D * Mh + D);
    return n + 2 * D;
}

void os_vit_destroy(os_ctx *ctx)
{
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v) return;
    if (v->buf) (void)hipFree(v->buf);
    rocblas_destroy_handle(v->blas);
    free(v);
    ctx->vit = nullptr;
}

// C[M x N] (row-major) = A[M x K] (row-major) . W[N x K]^T (row-major)
static int gemm_nt(os_ctx *ctx, os_vit_state *v, size_t M, int N, int K, const float *A, const float *W, float *Cm)
{
    const float alpha = 1.0f, beta = 0.0f;
    if (rocblas_sgemm(v->blas, rocblas_operation_transpose, rocblas_operation_none, N, (int)M, K, &alpha, W, K, A, K, &beta, Cm, N) !=
        rocblas_status_success)
        return os_fail(ctx, -20, "rocblas_sgemm failed (ViT)");
    return 0;
}

extern "C" {

size_t os_vit_param_count(const os_vit_dims *d)
{
    VitDims v{d->img_size, d->patch_size, d->embed_dim, d->depth, d->num_heads, d->mlp_hidden};
    return vit_param_count(v);
}

int os_vit_load(os_ctx *ctx, const os_vit_dims *d, const float *w_flat)
{
    OS_CHECK_CTX(ctx);
    if (!d || !w_flat) return os_fail(ctx, -2, "os_vit_load: null pointer");
    if (d->embed_dim > 256 || d->embed_dim % d->num_heads || d->img_size % d->patch_size || d->in_chans != 1)
        return os_fail(ctx, -4, "os_vit_load: unsupported dimensions (embed_dim <= 256, one input channel)");
    const int hd = d->embed_dim / d->num_heads;
    if (hd != 32 && hd != 64) return os_fail(ctx, -4, "os_vit_load: head dimension must be 32 or 64");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v) {
        v = (os_vit_state *)calloc(1, sizeof(os_vit_state));
        if (!v || rocblas_create_handle(&v->blas) != rocblas_status_success) { free(v); return os_fail(ctx, -13, "os_vit_load: no rocBLAS handle"); }
        ctx->vit = v;
    }
    v->d = VitDims{d->img_size, d->patch_size, d->embed_dim, d->depth, d->num_heads, d->mlp_hidden};
    v->w = w_flat;
    return 0;
}

int os_vit_encode(os_ctx *ctx, int32_t N, const float *images, float *latent, void *stream)
{
    OS_CHECK_CTX(ctx);
    os_vit_state *v = (os_vit_state *)ctx->vit;
    if (!v || !v->w) return os_fail(ctx, -5, "os_vit_encode: call os_vit_load first");
    if (N <= 0 || !images || !latent) return os_fail(ctx, -2, "os_vit_encode: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    rocblas_set_stream(v->blas, s);
    const VitDims &d = v->d;
    const int D = d.dim, L = ntok(d), G = grid_of(d), PP = d.patch * d.patch, Mh = d.mlp, hd = D / d.heads;
    const size_t M = (size_t)N * L, Mp = (size_t)N * G * G;
    // scratch: X [M][D] | Y [M][D] | tmp [M][D] | big [M][max(3D, mlp)] | patches [Mp][PP] | tok [Mp][D]
    const size_t bigw = (size_t)(3 * D > Mh ? 3 * D : Mh);
    const size_t need = M * D * 3 + M * bigw + Mp * PP + Mp * D;
    if (os_ensure_scratch(ctx, &v->buf, &v->buf_floats, need)) return -10;
    float *X = v->buf, *Y = X + M * D, *tmp = Y + M * D, *big = tmp + M * D;
    float *patches = big + M * bigw, *tok = patches + Mp * PP;
    // weights
    const float *w = v->w;
    const float *patch_w = w; w += (size_t)D * PP;
    const float *patch_b = w; w += D;
    const float *cls = w; w += D;
    const float *pos = w; w += (size_t)L * D;
    const int TB = 256, GB = 4096;
    hipLaunchKernelGGL(patchify_kernel, dim3(GB), dim3(TB), 0, s, N, d, images, patches);
    if (gemm_nt(ctx, v, Mp, D, PP, patches, patch_w, tok)) return -20;
    hipLaunchKernelGGL(assemble_kernel, dim3(GB), dim3(TB), 0, s, N, L, D, tok, patch_b, cls, pos, X);
    OS_HIP(ctx, hipGetLastError());
    for (int blk = 0; blk < d.depth; blk++) {
        const float *ln1w = w; w += D; const float *ln1b = w; w += D;
        const float *qkvw = w; w += (size_t)3 * D * D; const float *qkvb = w; w += 3 * D;
        const float *projw = w; w += (size_t)D * D; const float *projb = w; w += D;
        const float *ln2w = w; w += D; const float *ln2b = w; w += D;
        const float *fc1w = w; w += (size_t)Mh * D; const float *fc1b = w; w += Mh;
        const float *fc2w = w; w += (size_t)D * Mh; const float *fc2b = w; w += D;
        hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, M, D, X, ln1w, ln1b, Y, (size_t)D);
        if (gemm_nt(ctx, v, M, 3 * D, D, Y, qkvw, big)) return -20;
        const size_t alds = (size_t)2 * L * (hd + 1) * sizeof(float);
        const int ntl = (L + 31) / 32;
        if (hd == 32 && ntl == 7) {              // the reference's shape: 197 tokens, head_dim 32 -> matrix cores
            const size_t mlds = ((size_t)2 * L * 33 + 4 * 32 * 33) * sizeof(float);
            if (!v->att_attr_set) {
                OS_HIP(ctx, hipFuncSetAttribute((const void *)attention_mfma_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                v->att_attr_set = true;
            }
            hipLaunchKernelGGL(attention_mfma_kernel<7>, dim3(N * d.heads), dim3(256), mlds, s, L, D, d.heads, big, qkvb, Y);
        } else if (hd == 32) hipLaunchKernelGGL(attention_kernel<32>, dim3(N * d.heads), dim3(256), alds, s, L, D, d.heads, big, qkvb, Y);
        else hipLaunchKernelGGL(attention_kernel<64>, dim3(N * d.heads), dim3(256), alds, s, L, D, d.heads, big, qkvb, Y);
        if (gemm_nt(ctx, v, M, D, D, Y, projw, tmp)) return -20;
        hipLaunchKernelGGL(bias_residual_kernel, dim3(GB), dim3(TB), 0, s, M * D, D, tmp, projb, X);
        hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, M, D, X, ln2w, ln2b, Y, (size_t)D);
        if (gemm_nt(ctx, v, M, Mh, D, Y, fc1w, big)) return -20;
        hipLaunchKernelGGL(bias_gelu_kernel, dim3(GB), dim3(TB), 0, s, M * Mh, Mh, fc1b, big);
        if (gemm_nt(ctx, v, M, D, Mh, big, fc2w, tmp)) return -20;
        hipLaunchKernelGGL(bias_residual_kernel, dim3(GB), dim3(TB), 0, s, M * D, D, tmp, fc2b, X);
        OS_HIP(ctx, hipGetLastError());
    }
    const float *nw = w; w += D; const float *nb = w;
    hipLaunchKernelGGL(cls_head_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, L, D, X, nw, nb, latent);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
