// Micro-probe (development): semantics of the 64-bit DPP forms kf_dense_rows.hip relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/dpp64_probe.hip -o tools/micro/dpp64_probe && tools/micro/dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S>
__global__ void k(const double *in, double *mov, double *fm)
{
    const int l = threadIdx.x;
    double v = in[l], o, acc = 1000.0, m = 2.0;
    asm volatile("s_nop 4\nv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(S));
    asm volatile("s_nop 4\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(m), "n"(S));
    mov[l] = o; fm[l] = acc;
}
template <int S> int run(const double *din, double *dmov, double *dfm)
{
    double mov[64], fm[64];
    hipLaunchKernelGGL(k<S>, dim3(1), dim3(64), 0, 0, din, dmov, dfm);
    hipMemcpy(mov, dmov, sizeof(mov), hipMemcpyDeviceToHost); hipMemcpy(fm, dfm, sizeof(fm), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const double want = 100.0 + (l & ~15) + S;
        if (mov[l] != want || fm[l] != 1000.0 + 2.0 * want) bad++;
    }
    printf("row_newbcast:%d  mov lanes 0,1,17,40,63 = %g %g %g %g %g  fmac lane 17 = %g  mismatches %d\n", S, mov[0], mov[1], mov[17], mov[40], mov[63], fm[17], bad);
    return bad;
}
int main()
{
    double in[64]; for (int l = 0; l < 64; l++) in[l] = 100.0 + l;
    double *din, *dmov, *dfm;
    hipMalloc(&din, sizeof(in)); hipMalloc(&dmov, sizeof(in)); hipMalloc(&dfm, sizeof(in));
    hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice);
    int bad = run<0>(din, dmov, dfm) + run<1>(din, dmov, dfm) + run<5>(din, dmov, dfm) + run<8>(din, dmov, dfm) + run<11>(din, dmov, dfm) + run<15>(din, dmov, dfm);
    printf(bad ? "DPP64 row_newbcast: UNEXPECTED semantics\n" : "DPP64 row_newbcast: lane S of each 16-lane row, as assumed\n");
    return bad != 0;
}
