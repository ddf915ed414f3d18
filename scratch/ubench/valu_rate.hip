// Instruction-issue microbenchmark for gfx950: cycles per wave64 instruction for a few VALU opcodes at 1 / 2 / 4 / 8 waves per SIMD,
// plus the shader clock (s_memtime cycles against wall_clock64's constant 100 MHz counter).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, uint64_t* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    uint64_t w0 = threadIdx.x, w1 = threadIdx.x + 5;
    uint32_t u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 5 + 2, u3 = u0 * 7 + 3, u4 = u0 + 9, u5 = u0 + 11, u6 = u0 + 13, u7 = u0 + 17;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) {  // v_fma_f32, 8 independent chains
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) {  // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            } else if (OP == 2) {  // v_xor_b32
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n"
                             "v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 3) {  // v_log_f32
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                             "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 4) {  // v_alignbit_b32
                asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %4, 7\n"
                             "v_alignbit_b32 %4, %4, %5, 7\n v_alignbit_b32 %5, %5, %6, 7\n v_alignbit_b32 %6, %6, %7, 7\n v_alignbit_b32 %7, %7, %0, 7\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 5) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 6) {  // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %0\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            } else if (OP == 7) {  // v_sin_f32
                asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n"
                             "v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 9) {  // v_bitop3_b32 (a ^ b ^ c)
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96\n"
                             "v_bitop3_b32 %4, %4, %5, %6 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x96\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0x96\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x96\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 10) {  // v_mad_u64_u32 (Philox's multiply)
                asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %3, s[10:11], %2, %1, %3\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %3, s[10:11], %2, %1, %3\n"
                             "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %3, s[10:11], %2, %1, %3\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %3, s[10:11], %2, %1, %3\n"
                             : "+v"(w0), "+v"(u1), "+v"(u2), "+v"(w1) : : "s10", "s11");
            } else if (OP == 11) {  // v_fma_f32, ONE dependent chain
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n" : "+v"(a0));
            } else if (OP == 12) {  // v_log_f32, ONE dependent chain
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %0, %0\n v_log_f32 %0, %0\n v_log_f32 %0, %0\n"
                             "v_log_f32 %0, %0\n v_log_f32 %0, %0\n v_log_f32 %0, %0\n v_log_f32 %0, %0\n" : "+v"(a0));
            } else if (OP == 13) {  // log -> mul -> sqrt -> mul, two dependent chains (the draw's radius chain)
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n"
                             "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n" : "+v"(a0), "+v"(a1) : "v"(a2));
            } else if (OP == 14) {  // v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 15) {  // v_xor_b32, ONE dependent chain
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                             "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n" : "+v"(u0) : "v"(u1));
            } else if (OP == 16) {  // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %0\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            } else if (OP == 17) {  // v_lshlrev_b32
                asm volatile("v_lshlrev_b32 %0, 9, %1\n v_lshlrev_b32 %1, 9, %2\n v_lshlrev_b32 %2, 9, %3\n v_lshlrev_b32 %3, 9, %4\n"
                             "v_lshlrev_b32 %4, 9, %5\n v_lshlrev_b32 %5, 9, %6\n v_lshlrev_b32 %6, 9, %7\n v_lshlrev_b32 %7, 9, %0\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 18) {  // v_sin + v_cos of the same operand, then a plain mul each (the draw's angle part), 4 chains
                asm volatile("v_sin_f32 %4, %0\n v_cos_f32 %5, %0\n v_sin_f32 %6, %1\n v_cos_f32 %7, %1\n"
                             "v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %6, %7\n v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 8) {  // v_mul_f32 + v_add_f32 mix
                asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n"
                             "v_mul_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_add_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                                          (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(w0 + w1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char* name) {
    float* out; uint64_t* cyc;
    const int maxb = 256 * 8;
    CHECK(hipMalloc(&out, maxb * 256 * 4)); CHECK(hipMalloc(&cyc, maxb * 8));
    const int iters = 4000;  // 4000 * 64 instructions per wave
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 100);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t h[maxb]; CHECK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double instr = (double)iters * 64;
        // s_memtime counts at a constant 100 MHz on gfx9+: cycles here are "refclk ticks"; derive ns per instruction from wall time
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD slot -> %.2f SIMD-ns per instr (x waves), memtime ticks %.0f\n", name, wps, ms,
               ms * 1e6 / instr, ms * 1e6 / instr / wps, avg);
    }
    return 0;
}

int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<6>("v_pk_add_f32"); run<8>("mul/add f32"); run<2>("v_xor_b32"); run<5>("v_add_u32"); run<4>("v_alignbit");
    run<3>("v_log_f32"); run<7>("v_sin_f32"); run<14>("v_sqrt_f32");
    run<9>("v_bitop3_b32"); run<17>("v_lshlrev_b32"); run<16>("v_pk_mul_f32"); run<10>("v_mad_u64_u32");
    run<11>("fma dep chain"); run<15>("xor dep chain"); run<12>("log dep chain"); run<13>("log-mul-sqrt-mul x2"); run<18>("sin+cos-mul-add x2");
    return 0;
}
