// Does a VALU instruction cost less when only part of the wave is active?  v_fma_f32 under EXEC masks of different shapes:
// all 64 lanes, the low 32, the low 16, one lane, every other lane, one lane in each half.  (Motivation: branches that run for
// 1-13 of 64 lanes -- primary-ray generation, the Dielectric branch -- priced at their instruction count predict gains that
// gating and batching them never delivered, profiles/r04_gen_ahead_experiment.txt.)
// hipcc --offload-arch=gfx950 -O3 -o exec_mask_rates exec_mask_rates.hip && ./exec_mask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 4096, kChains = 8;

__global__ void __launch_bounds__(256) bench(uint32_t* out, uint64_t mask, float seed) {
    float f[kChains];
    for (int c = 0; c < kChains; ++c) f[c] = seed + (float)threadIdx.x + (float)c;
    const uint32_t lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < kIters; ++i) {
#pragma unroll
            for (int c = 0; c < kChains; ++c) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c]));
        }
    }
    float acc = 0.0f;
    for (int c = 0; c < kChains; ++c) acc += f[c];
    if (acc == 12345.678f) out[0] = 1u;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char* name; uint64_t mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"low 32 lanes", 0xffffffffull}, {"high 32 lanes", 0xffffffff00000000ull}, {"low 16 lanes", 0xffffull},
        {"lanes 16-31", 0xffff0000ull}, {"one lane (0)", 1ull}, {"one lane (40)", 1ull << 40}, {"every other lane", 0x5555555555555555ull},
        {"lane 0 + lane 32", 1ull | (1ull << 32)}, {"lane 0 + lane 16", 1ull | (1ull << 16)}, {"13 scattered lanes", 0x0101101001011011ull},
    };
    for (auto& c : cases) {
        const int blocks = 256 * 8; // 8 waves per SIMD
        bench<<<blocks, 256>>>(d, c.mask, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        bench<<<blocks, 256>>>(d, c.mask, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = 8.0 * kIters * kChains;
        printf("%-20s %8.3f ms  %5.2f cycles per wave-instruction\n", c.name, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
