// Which property of a kernel brings the one-off 50-90 ms queue stall to a fresh process (round 5)?  Each mode launches a small kernel in
// bursts of 25 for SECS seconds after 100 warm-up launches and prints every burst over 5 ms:
//   dyn<KB>   dynamic LDS of that size (hipFuncSetAttribute once)      static100  100 KB of static LDS
//   sincospi  fp64 sincospi table like the general-size kernels        plain      nothing special
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_dyn(float* out, int n) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = i * 0.5f;
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[(threadIdx.x * 7) % n];
}
__global__ void k_static(float* out) {
    __shared__ float lds[25600];
    for (int i = threadIdx.x; i < 25600; i += blockDim.x) lds[i] = i * 0.5f;
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[(threadIdx.x * 7) % 25600];
}
__global__ void k_sincospi(float* out) {
    double sn, cs;
    sincospi(2.0 * (double)threadIdx.x / 152.0, &sn, &cs);
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(sn + cs);
}
__global__ void k_plain(float* out) { out[blockIdx.x * blockDim.x + threadIdx.x] = threadIdx.x; }
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "plain";
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; CHECK(hipMalloc(&out, 256 * 512 * 4));
    int kb = 0;
    if (!strncmp(mode, "dyn", 3)) { kb = atoi(mode + 3); CHECK(hipFuncSetAttribute((const void*)k_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    auto launch = [&]() {
        if (kb) hipLaunchKernelGGL(k_dyn, dim3(256), dim3(512), kb * 1024, 0, out, kb * 256);
        else if (!strcmp(mode, "static100")) hipLaunchKernelGGL(k_static, dim3(256), dim3(512), 0, 0, out);
        else if (!strcmp(mode, "sincospi")) hipLaunchKernelGGL(k_sincospi, dim3(256), dim3(512), 0, 0, out);
        else hipLaunchKernelGGL(k_plain, dim3(256), dim3(512), 0, 0, out);
    };
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t00 = now();
    for (int i = 0; i < 100; ++i) launch();
    CHECK(hipDeviceSynchronize());
    const double t0 = now();
    int bursts = 0, slow = 0;
    while (now() - t0 < secs) {
        const double a = now();
        for (int i = 0; i < 25; ++i) launch();
        const double b = now();
        CHECK(hipDeviceSynchronize());
        const double c = now();
        ++bursts;
        if (c - a > 5e-3) { ++slow; printf("   %s: burst %d at +%.2f s: host %.1f ms, sync wait %.1f ms\n", mode, bursts, a - t00, (b - a) * 1e3, (c - b) * 1e3); }
    }
    printf("%s: %d bursts, %d over 5 ms\n", mode, bursts, slow);
    return 0;
}
