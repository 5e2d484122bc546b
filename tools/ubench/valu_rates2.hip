// Issue cost of selects, compares and cross-lane instructions next to v_fma_f32 (follow-up of valu_rates.hip).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip && ./valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 2048, kChains = 8;

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t* out, uint32_t seed) {
    uint32_t v[kChains], w[kChains];
    float f[kChains];
    for (int c = 0; c < kChains; ++c) {
        v[c] = seed + threadIdx.x * 7u + c;
        w[c] = v[c] * 3u + 1u;
        f[c] = (float)v[c];
    }
    unsigned long long mask = 0x5555aaaa5555aaaaull ^ seed;
    asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c]));
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(w[c]));                 // mask in vcc, never rewritten
            if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(w[c]), "s"(mask));      // mask in an SGPR pair
            if (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]) : "vcc"); // compare + select pair
            if (OP == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(f[c]), "v"(f[(c + 1) % kChains]) : "vcc");
            if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]));
            if (OP == 6) asm volatile("v_bfi_b32 %0, %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 7) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 8) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 9) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_add_u32 %0, s20, %0" : "+v"(v[c]) : : "s20");
            if (OP == 10) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]));
            if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[c]) : "v"(w[c]));
            if (OP == 14) asm volatile("v_cvt_f32_u32 %0, %1" : "+v"(f[c]) : "v"(v[c]));
            if (OP == 15) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]) : "vcc");
            if (OP == 16) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]));
            if (OP == 17) asm volatile("v_div_fmas_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]) : "vcc");
        }
    }
    uint32_t acc = 0;
    for (int c = 0; c < kChains; ++c) acc ^= v[c] ^ __float_as_uint(f[c]);
    if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
void run(const char* name, int per_op) {
    uint32_t* d;
    (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    bench<OP><<<blocks, 256>>>(d, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    bench<OP><<<blocks, 256>>>(d, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 8.0 * kIters * kChains;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f cycles per statement (%d instruction(s))\n", name, ms, cycles / per_simd, per_op);
    (void)hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_cndmask_b32 (vcc)", 1);
    run<2>("v_cndmask_b32_e64 (sgpr)", 1);
    run<3>("v_cmp + v_cndmask", 2);
    run<4>("v_cmp_lt_f32 -> vcc", 1);
    run<5>("v_max_f32", 1);
    run<6>("v_bfi_b32", 1);
    run<7>("v_mov_b32_dpp row_shr", 1);
    run<8>("ds_bpermute_b32 + wait", 1);
    run<9>("v_readlane + v_add", 2);
    run<10>("v_mbcnt_lo", 1);
    run<11>("v_add_f32", 1);
    run<12>("v_add_u32", 1);
    run<13>("v_lshl_add_u32", 1);
    run<14>("v_cvt_f32_u32", 1);
    run<15>("v_div_scale_f32", 1);
    run<16>("v_div_fixup_f32", 1);
    run<17>("v_div_fmas_f32", 1);
    return 0;
}
