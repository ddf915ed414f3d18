// standalone timing harness for the power-noise kernels (variants via -D flags)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../include/sonar_hip.h"
int main(int argc, char** argv) {
    const int64_t B = 512, planes = B * 4, H = 128, W = 128;
    float *filt, *out, *z; double* part;
    hipMalloc(&filt, H * 65 * 4); hipMalloc(&out, planes * H * W * 4); hipMalloc(&z, planes * H * 65 * 8); hipMalloc(&part, 2048 * 16);
    float* hf = (float*)malloc(H * 65 * 4); for (int i = 0; i < H * 65; ++i) hf[i] = 0.5f + (i % 7) * 0.1f;
    hipMemcpy(filt, hf, H * 65 * 4, hipMemcpyHostToDevice);
    sonar_power_spectrum_f32(z, planes, H, W, 1, 0, 0, nullptr);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(a); for (int i = 0; i < 10; ++i) fn(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-22s %8.1f us\n", name, ms / 10 * 1000);
    };
    timeit("replay", [&] { sonar_power_irfft2_f32(z, filt, out, planes, H, W, 1, 0, 0, nullptr, nullptr); });
    timeit("gen", [&] { sonar_power_irfft2_f32(nullptr, filt, out, planes, H, W, 1, 0, 0, nullptr, nullptr); });
    timeit("gen+stats", [&] { sonar_power_irfft2_f32(nullptr, filt, out, planes, H, W, 1, 0, 0, part, nullptr); });
    timeit("fused normalised", [&] { sonar_power_noise_f32(filt, out, planes, H, W, 1, 0, 0, 1.0f, 2.5f, part, nullptr); });
    timeit("spectrum dump", [&] { sonar_power_spectrum_f32(z, planes, H, W, 1, 0, 0, nullptr); });
    printf("err: %s\n", sonar_last_error());
    return 0;
}
