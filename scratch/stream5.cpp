// microbench: 3 reads + 2 writes of float4 streams (the Euler momentum step's traffic), variants of grid / unroll / nontemporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k5(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                         float4* __restrict__ o1, float4* __restrict__ o2, long nv) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nv; i += UNROLL * stride) {
        float4 va[UNROLL], vb[UNROLL], vc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { va[u] = a[i + u * stride]; vb[u] = b[i + u * stride]; vc[u] = c[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float4 r1, r2;
            r1.x = va[u].x * 0.5f + vb[u].x; r1.y = va[u].y * 0.5f + vb[u].y; r1.z = va[u].z * 0.5f + vb[u].z; r1.w = va[u].w * 0.5f + vb[u].w;
            r2.x = vc[u].x * 0.9f + r1.x; r2.y = vc[u].y * 0.9f + r1.y; r2.z = vc[u].z * 0.9f + r1.z; r2.w = vc[u].w * 0.9f + r1.w;
            if (NT) {
                __builtin_nontemporal_store(r1.x, &o1[i + u * stride].x); __builtin_nontemporal_store(r1.y, &o1[i + u * stride].y);
                __builtin_nontemporal_store(r1.z, &o1[i + u * stride].z); __builtin_nontemporal_store(r1.w, &o1[i + u * stride].w);
                __builtin_nontemporal_store(r2.x, &o2[i + u * stride].x); __builtin_nontemporal_store(r2.y, &o2[i + u * stride].y);
                __builtin_nontemporal_store(r2.z, &o2[i + u * stride].z); __builtin_nontemporal_store(r2.w, &o2[i + u * stride].w);
            } else { o1[i + u * stride] = r1; o2[i + u * stride] = r2; }
        }
    }
    for (; i < nv; i += stride) { float4 va = a[i], vb = b[i], vc = c[i]; float4 r1 = va, r2 = vc; r1.x += vb.x; r2.x += r1.x; o1[i] = r1; o2[i] = r2; }
}

template <int UNROLL, bool NT>
int run(const char* name, int grid, float4** buf, long nv) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k5<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], nv);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k5<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], nv);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 10 < best) best = ms / 10;
    }
    printf("%-28s grid %6d: %7.1f us  %6.2f TB/s\n", name, grid, best * 1e3, 5.0 * nv * 16 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const long n = 512L * 4 * 128 * 128, nv = n / 4;
    float4* buf[5];
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
    for (int grid : {2048, 4096, 8192, 16384, 32768}) {
        run<1, false>("unroll1", grid, buf, nv);
        run<2, false>("unroll2", grid, buf, nv);
        run<4, false>("unroll4", grid, buf, nv);
        run<2, true>("unroll2 nontemporal", grid, buf, nv);
        run<4, true>("unroll4 nontemporal", grid, buf, nv);
    }
    return 0;
}
