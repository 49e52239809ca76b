// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS code base's access pattern: coalesced one-dword-per-lane
// loads and stores (what the Kalman/fused kernels issue), on a buffer larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void rd_dword(const float *src, float *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += stride) s += src[i];
    if (s == 123.456f) dst[0] = s;
}
__global__ void wr_dword(float *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)i;
}
int main()
{
    const size_t n = (size_t)1 << 28;   // 1 GiB of floats
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4);
    hipLaunchKernelGGL(rd_dword, dim3(2048), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(wr_dword, dim3(2048), dim3(256), 0, 0, b, n);
    hipDeviceSynchronize();
    printf("bytes read by rd_dword: %zu  bytes written by wr_dword: %zu\n", n * 4, n * 4);
    return 0;
}
