// Micro-benchmark (development): cycles per v_mfma_f32_32x32x16_bf16 for the instruction mix of gru_layer_bf16_kernel -- one wave per
// SIMD (256-thread workgroups, 512 registers), accumulators in AGPRs or VGPRs, 1 / 3 / 6-deep dependent chains, with and without
// ds_read_b32 / buffer_load_dwordx4 / VALU fillers between the MFMAs -- every CU busy (the clock the chip holds under this load is
// part of the answer: wall time is printed beside the cycle count).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_bf16_rate tools/micro/mfma_bf16_rate.hip && tools/micro/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define MF_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MF_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// MODE 0: AGPR acc, chain 1 (12 accumulators round-robin); 1: VGPR acc; 2: AGPR acc, 6-deep chains; 3: mode 0 + 3 VALU per MFMA;
// 4: mode 0 + 40 ds_read_b32 per 72; 5: mode 0 + 9 buffer_load_dwordx4 per 72; 6: everything (3 + 4 + 5)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const int *w, int iters, unsigned long long *out, float *sink)
{
    __shared__ float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    i32x4 a0 = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, b0 = a0, a1 = a0, b1 = a0;
    float v0 = lane, v1 = lane + 1, v2 = lane + 2;
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(w), 0, 1 << 20, 0x00020000);
    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)lds + lane * 4;
    i32x4 wl[9];
    float dl[8];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int j = 0; j < 9; j++) wl[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (j + (it & 15) * 9) * 1024, 0));
        }
#pragma unroll
        for (int m = 0; m < 72; m++) {
            const int ai = (MODE == 2) ? (m / 6) % 12 : m % 12;
            if (MODE == 1) MF_V(acc[ai], (m & 1) ? a1 : a0, (m & 2) ? b1 : b0);
            else MF_A(acc[ai], (m & 1) ? a1 : a0, (m & 2) ? b1 : b0);
            if (MODE == 3 || MODE == 6) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v2) : "v"(v0), "v"(v1));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v1) : "v"(v2));
            }
            if ((MODE == 4 || MODE == 6) && m < 40) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dl[m & 7]) : "v"(la), "n"((m * 256) % 16384));
        }
        if (MODE == 4 || MODE == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dl[0]), "+v"(dl[1]), "+v"(dl[2]), "+v"(dl[3]), "+v"(dl[4]), "+v"(dl[5]), "+v"(dl[6]), "+v"(dl[7]));
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int j = 0; j < 9; j++) a0[j & 3] ^= wl[j][0] & 0;        // consume (keeps the loads)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2;
#pragma unroll
    for (int i = 0; i < 12; i++) s += acc[i][0];
    if (MODE == 4 || MODE == 6) s += dl[0] + dl[7];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(const char *what, const int *w, unsigned long long *out, float *sink)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, 200, out, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, iters, out, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("%-72s %6.2f cycles per MFMA | %7.3f ms | %5.2f GHz implied\n", what, (double)c / (72.0 * iters), ms, (double)c / (ms * 1e6));
}

int main()
{
    int *w; unsigned long long *out; float *sink;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, 8); hipMalloc(&sink, 4);
    run<0>("AGPR accumulators, 12 round-robin", w, out, sink);
    run<1>("VGPR accumulators, 12 round-robin", w, out, sink);
    run<2>("AGPR accumulators, 6-deep dependent chains", w, out, sink);
    run<3>("AGPR, + 3 VALU per MFMA", w, out, sink);
    run<4>("AGPR, + 40 ds_read_b32 per 72 MFMAs", w, out, sink);
    run<5>("AGPR, + 9 buffer_load_dwordx4 (L2) per 72 MFMAs", w, out, sink);
    run<6>("AGPR, + VALU + ds_read + buffer_load", w, out, sink);
    return 0;
}
