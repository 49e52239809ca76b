// Issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU instruction kinds the Kalman part uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, unsigned long long *cyc)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x * 0.001f + 1.f;
    f2 d[8];
    for (int i = 0; i < 8; i++) d[i] = (f2){a + i, a - i};
    const f2 m = (f2){1.0001f, 0.9999f}, c = (f2){0.5f, 0.25f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int v = 0; v < 8; v++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d[v].x) : "v"(m.x), "v"(c.x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[v]) : "v"(m), "v"(c));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[v]) : "v"(m));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(d[v].x));
                if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(d[v].x));
                if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[v].x) : "v"(d[(v + 1) & 7].y));
                if (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d[v].x), "+v"(d[v].y));
                if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[v]) : "v"(m), "v"(c));
                if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[v]) : "v"(m));
                if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[v].x) : "v"(m.x));
                if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d[v].x) : "s"(1.0001f), "v"(c.x));
                if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(d[v]) : "v"(m), "v"(c));
                if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d[v].x) : "v"(m.x));
                if (KIND == 14) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[v]) : "v"(m));
                if (KIND == 15) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[v]) : "v"(m), "v"(c));
                if (KIND == 16) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[v]));
                if (KIND == 17) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[v]) : "v"(d[(v + 2) & 7]));
                if (KIND == 18) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[v]) : "v"(d[(v + 4) & 7]), "v"(m));
                if (KIND == 13) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(d[v].x));
                if (KIND == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(d[v].x) : "v"(m.x) : "s20", "s21");
                if (KIND == 20) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(d[v].x) : "v"(m.x), "v"(c.x));
                if (KIND == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(d[v].x) : "v"(m.x));
                if (KIND == 22) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(d[v].x), "v"(m.x) : "vcc");
                if (KIND == 23) asm volatile("v_cmp_gt_f32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc" : "+v"(d[v].x) : "v"(m.x) : "vcc");
                if (KIND == 30) asm volatile("v_cmp_gt_f32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %2, vcc\nv_cndmask_b32 %1, %1, %2, vcc" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "vcc");
                if (KIND == 31) asm volatile("v_cmp_gt_f32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %2, vcc\nv_cndmask_b32 %1, %1, %2, vcc\nv_cndmask_b32 %0, %0, %2, vcc\nv_cndmask_b32 %1, %1, %2, vcc" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "vcc");
                if (KIND == 32) asm volatile("v_cmp_gt_f32 s[20:21], %0, %2\nv_cndmask_b32_e64 %0, %0, %2, s[20:21]\nv_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "s20", "s21");
                if (KIND == 33) asm volatile("v_cmp_gt_f32 vcc, %0, %2\nv_mul_f32 %1, %1, %2\nv_mul_f32 %1, %1, %2\nv_cndmask_b32 %0, %0, %2, vcc\nv_cndmask_b32 %1, %1, %2, vcc" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "vcc");
                if (KIND == 34) asm volatile("v_cmp_gt_f32 vcc, %0, %2\ns_nop 1\nv_cndmask_b32 %0, %0, %2, vcc\nv_cndmask_b32 %1, %1, %2, vcc" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "vcc");
                if (KIND == 35) asm volatile("v_cmp_gt_f32 s[20:21], %0, %2\ns_nop 1\nv_cndmask_b32_e64 %0, %0, %2, s[20:21]\nv_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "s20", "s21");
                if (KIND == 36) asm volatile("v_cmp_gt_f32 vcc, %0, %2\ns_nop 1\nv_cndmask_b32 %0, %0, %2, vcc\nv_mul_f32 %1, %1, %2\nv_cndmask_b32 %1, %1, %2, vcc" : "+v"(d[v].x), "+v"(d[v].y) : "v"(m.x) : "vcc");
                if (KIND == 24) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(d[v].x) : "s20");
                if (KIND == 25) asm volatile("v_writelane_b32 %0, s20, 3" : "+v"(d[v].x) : : "s20");
                if (KIND == 26) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[v]) : "v"(m));
                if (KIND == 27) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d[v].x) : "v"(m.x), "v"(c.x));
                if (KIND == 28) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[v]) : "v"(m));
                if (KIND == 29) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[v]) : "v"(m));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += d[i].x + d[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run1(float *d, const char *name, int per, int threads)
{
    unsigned long long *cyc; hipHostMalloc(&cyc, 8); *cyc = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, d, 2000, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, d, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 32 * per;
    printf("%-28s %d waves/SIMD: %.3f ms, s_memtime %.2f ticks/instr/wave (clock ratio %.3f GHz-equivalent)\n", name, threads / 256, ms,
           (double)*cyc / n, (double)*cyc / (ms * 1e6));
}
template <int KIND> void run(float *d, const char *name, int per) { run1<KIND>(d, name, per, 256); run1<KIND>(d, name, per, 512); }
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<0>(d, "v_fma_f32", 1);
    run<10>(d, "v_fma_f32 (sgpr src)", 1);
    run<9>(d, "v_mul_f32", 1);
    run<1>(d, "v_pk_fma_f32", 1);
    run<11>(d, "v_pk_fma_f32 op_sel bcast", 1);
    run<2>(d, "v_pk_mul_f32", 1);
    run<8>(d, "v_pk_add_f32", 1);
    run<3>(d, "v_exp_f32", 1);
    run<4>(d, "v_rcp_f32", 1);
    run<5>(d, "v_mov_b32_dpp row_share", 1);
    run<6>(d, "v_permlane32_swap_b32", 1);
    run<7>(d, "v_fma_f64", 1);
    run<15>(d, "v_fmac_f64", 1);
    run<14>(d, "v_fmac_f64_dpp (self)", 1);
    run<18>(d, "v_fmac_f64_dpp (other)", 1);
    run<17>(d, "v_mov_b64_dpp", 1);
    run<16>(d, "v_rcp_f64", 1);
    run<12>(d, "v_cndmask_b32", 1);
    run<13>(d, "accvgpr write+read", 2);
    run<19>(d, "v_cndmask_b32_e64 (sgpr mask)", 1);
    run<27>(d, "v_cndmask_b32 (no self dep)", 1);
    run<20>(d, "v_bfi_b32", 1);
    run<21>(d, "v_and_b32", 1);
    run<22>(d, "v_cmp_gt_f32 vcc", 1);
    run<23>(d, "v_cmp + v_cndmask", 2);
    run<30>(d, "v_cmp + 2 v_cndmask", 3);
    run<31>(d, "v_cmp + 4 v_cndmask", 5);
    run<32>(d, "v_cmp sgpr + 2 v_cndmask_e64", 3);
    run<33>(d, "v_cmp + 2 v_mul + 2 v_cndmask", 5);
    run<34>(d, "v_cmp + nop + 2 v_cndmask", 3);
    run<35>(d, "v_cmp sgpr + nop + 2 cndmask_e64", 3);
    run<36>(d, "v_cmp + nop + cnd + mul + cnd", 4);
    run<24>(d, "v_readlane_b32", 1);
    run<25>(d, "v_writelane_b32", 1);
    run<26>(d, "v_max_f64", 1);
    run<28>(d, "v_mul_f64", 1);
    run<29>(d, "v_add_f64", 1);
    return 0;
}
