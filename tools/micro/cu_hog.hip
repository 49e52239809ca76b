// cu_hog.hip -- occupies `nblocks` compute units for `ms` milliseconds with ONE long kernel: every workgroup claims 150 KB of LDS (one
// workgroup per CU, nothing that needs more than 10 KB of LDS fits beside it) and spins on the wall clock.  Workgroup i lands on XCD
// i % 8, so 240 workgroups leave two free CUs on every XCD of an MI355X.  tests/test_gpu_contention.py builds it on the GPU box
// (hipcc --offload-arch=gfx950 tools/micro/cu_hog.hip -o cu_hog) and runs it beside the layer-pipelined GRU launches.
// usage: cu_hog <nblocks> <ms>     prints "hog running" once the kernel has been submitted, "hog done" when it has finished
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(64) void hog_kernel(unsigned long long ticks, int *sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) {
        // (mostly asleep: the clock read travels over a path the whole chip shares; a tight loop of 240 x 64 lanes reading it slowed
        // OTHER kernels' cache-bypassing polls eightfold)
        for (int i = 0; i < 64; i++) { acc += lds[(threadIdx.x + acc) & 63]; __builtin_amdgcn_s_sleep(64); }
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int nblocks = argc > 1 ? atoi(argv[1]) : 240;
    const double ms = argc > 2 ? atof(argv[2]) : 3000.0;
    int rate_khz = 100000;                                   // wall_clock64 ticks at 100 MHz on gfx9
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    int *sink = nullptr;
    if (hipMalloc((void **)&sink, 4) != hipSuccess) return 2;
    if (hipFuncSetAttribute((const void *)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return 3;
    hipLaunchKernelGGL(hog_kernel, dim3(nblocks), dim3(64), 150 * 1024, 0, (unsigned long long)(ms * rate_khz), sink);
    if (hipGetLastError() != hipSuccess) return 4;
    printf("hog running\n"); fflush(stdout);
    if (hipDeviceSynchronize() != hipSuccess) return 5;
    printf("hog done\n");
    return 0;
}
