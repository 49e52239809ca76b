// How fast does one wave stream v_mfma_f32_32x32x2_f32 when the B operand of every MFMA comes from a buffer_load
// (development aid)?  MODE 0: B from a register; 1: loads issued, results unused; 2: loads feed the MFMAs (ring of 8 x 3);
// 3: B from LDS (ds_read_b32 ring).  One or two waves per SIMD.  Table is 6 KB (L1-resident) or 393 KB (L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0));
}
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(float *out, const float *tab, uint32_t tab_bytes, int iters, unsigned long long *cyc)
{
    __shared__ float lds[64 * 24 * 4];
    for (int i = threadIdx.x; i < 64 * 24 * 4; i += NT) lds[i] = 1.0f;
    __syncthreads();
    f32x16 acc[3];
    for (int i = 0; i < 3; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab), 0, tab_bytes, 0x00020000);
    const uint32_t wl = lane * 4u;
    float a = lane * 0.001f, wb[8][3], sink = 0.f;
    uint32_t base = (uint32_t)wave * 49152u % tab_bytes;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int g = 0; g < 3; g++) wb[j][g] = MODE == 3 ? lds[(j * 3 + g) * 64 + lane] : buf_load(r, wl + (j * 3 + g) * 256u, base);
    unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
        const uint32_t so = __builtin_amdgcn_readfirstlane((base + (uint32_t)(it + 1) * 6144u) % tab_bytes);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 2 || MODE == 3) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j][2], acc[2], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
                if (MODE == 1) sink += wb[j][0] + wb[j][1] + wb[j][2];
            }
            if (MODE == 3) {
#pragma unroll
                for (int g = 0; g < 3; g++) wb[j][g] = lds[(j * 3 + g) * 64 + lane + (it & 3) * 1536];
            } else if (MODE >= 1) {
#pragma unroll
                for (int g = 0; g < 3; g++) wb[j][g] = buf_load(r, wl + (j * 3 + g) * 256u, so);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long c1 = clock64();
    float s = sink;
    for (int i = 0; i < 3; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    for (int j = 0; j < 8; j++) for (int g = 0; g < 3; g++) s += wb[j][g];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <int MODE, int NT> void run(const char *name, float *d, const float *tab, uint32_t bytes, unsigned long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, d, tab, bytes, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, d, tab, bytes, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD %d table %7u B: %.3f ms, %.1f cycles per MFMA of a wave (%.1f per SIMD slot)\n", name, NT / 256, bytes, ms,
           (double)cyc[0] / (iters * 24.0), (double)cyc[0] / (iters * 24.0) / (NT / 256));
}
int main()
{
    float *d, *tab; unsigned long long *cyc;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&tab, 1 << 20); hipMemset(tab, 0, 1 << 20);
    hipHostMalloc((void **)&cyc, 64);
    for (uint32_t bytes : {6144u, 393216u}) {
        run<0, 256>("B from a register", d, tab, bytes, cyc);
        run<1, 256>("loads issued, unused by MFMA", d, tab, bytes, cyc);
        run<2, 256>("loads feed the MFMAs", d, tab, bytes, cyc);
        run<3, 256>("ds_read feeds the MFMAs", d, tab, bytes, cyc);
        run<0, 512>("B from a register", d, tab, bytes, cyc);
        run<1, 512>("loads issued, unused by MFMA", d, tab, bytes, cyc);
        run<2, 512>("loads feed the MFMAs", d, tab, bytes, cyc);
        run<3, 512>("ds_read feeds the MFMAs", d, tab, bytes, cyc);
    }
    return 0;
}
