// Issue cost of the integer multiplies Philox4x32 is made of, next to v_fma_f32 (2 cycles per wave64 on a SIMD-32).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 2048, kChains = 8;

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t* out, uint32_t seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint32_t v[kChains];
    float f[kChains];
    f2 p[kChains];
    for (int c = 0; c < kChains; ++c) {
        v[c] = seed + threadIdx.x * 7u + c;
        f[c] = (float)v[c];
        p[c] = f2{f[c], f[c] + 1.0f};
    }
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v[c]));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(v[c]));
            if (OP == 3) {
                uint64_t r;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "=v"(r) : "v"(v[c]) : "vcc");
                v[c] = (uint32_t)(r >> 32) ^ (uint32_t)r;
            }
            if (OP == 4) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(v[c]));
            if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(v[c]));
            if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
            if (OP == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
            if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[c]));
            if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[c]));
            if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[c]));
            if (OP == 11) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(f[c]));
            if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(v[c]));
            if (OP == 13) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[c]));
            if (OP == 14) asm volatile("v_mul_hi_u32_u24 %0, %0, %0" : "+v"(v[c]));
            if (OP == 15) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(*(double*)&p[c]));
        }
    }
    uint32_t acc = 0;
    for (int c = 0; c < kChains; ++c) acc ^= v[c] ^ __float_as_uint(f[c]) ^ __float_as_uint(p[c].x) ^ __float_as_uint(p[c].y);
    if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
void run(const char* name, int extra_per_op) {
    uint32_t* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    bench<OP><<<blocks, 256>>>(d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<OP><<<blocks, 256>>>(d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 8 waves * kIters * kChains (* ops per chain step)
    const double per_simd = 8.0 * kIters * kChains;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction (%d extra VALU per op not counted)\n", name, ms, cycles / per_simd, extra_per_op);
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 0);
    run<1>("v_mul_lo_u32", 0);
    run<2>("v_mul_hi_u32", 0);
    run<3>("v_mad_u64_u32", 1);
    run<4>("v_xor_b32", 0);
    run<5>("v_mul_u32_u24", 0);
    run<6>("v_rcp_f32", 0);
    run<7>("v_sqrt_f32", 0);
    run<8>("v_pk_fma_f32", 0);
    run<9>("v_pk_mul_f32", 0);
    run<10>("v_pk_add_f32", 0);
    run<11>("v_max3_f32", 0);
    run<12>("v_cndmask_b32", 0);
    run<13>("v_mul_f32", 0);
    run<14>("v_mul_hi_u32_u24", 0);
    run<15>("v_fma_f64", 0);
    return 0;
}
