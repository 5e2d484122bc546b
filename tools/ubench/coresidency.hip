// coresidency.hip -- which resource of a persistent kernel keeps a small kernel on another stream from becoming resident?
// A grid of 512 spinning workgroups (2 per CU) with a chosen size, VGPR / SGPR footprint and dynamic LDS runs for ~20 ms on
// stream A; 3 ms in, a kernel of 64 workgroups x 256 threads (16 VGPRs, no LDS) is launched on stream B.  Printed: how long
// after its launch the small kernel finished.  Co-resident: tens of microseconds.  Blocked: the rest of the spin.
// build: hipcc --offload-arch=gfx950 -O2 -o coresidency coresidency.hip ; run: ./coresidency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

template <int VGPR, int SGPR>
__global__ void __launch_bounds__(1024) spin(unsigned long long ticks, float* out) {
    extern __shared__ float lds[];
    if (VGPR >= 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VGPR >= 64 && VGPR < 80) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (SGPR >= 100) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    const unsigned long long t0 = wall_clock64();
    float acc = 0.0f;
    while (wall_clock64() - t0 < ticks) acc += 1.0f;
    if (acc < 0.0f) out[threadIdx.x] = acc + lds[0];
}
__global__ void __launch_bounds__(256) small(float* out) { out[blockIdx.x * 256 + threadIdx.x] = 1.0f; }

template <int VGPR, int SGPR>
static void run(int threads, int lds, int small_threads) {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    float* out;
    hipMalloc(&out, 1 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin<VGPR, SGPR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spin<VGPR, SGPR>, threads, lds);
    const unsigned long long ticks = 2000000ull; // 20 ms at 100 MHz
    hipLaunchKernelGGL((spin<VGPR, SGPR>), dim3(256 * per_cu), dim3(threads), lds, a, ticks, out);
    std::this_thread::sleep_for(std::chrono::milliseconds(3));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(small, dim3(64), dim3(small_threads), 0, b, out);
    hipStreamSynchronize(b);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(a);
    printf("spin: %4d threads x %d per CU = %2d waves/CU, %2d VGPRs, %s SGPRs, %6d B LDS | small kernel of %3d-thread groups done after %7.3f ms  %s\n", threads, per_cu,
           threads / 64 * per_cu, VGPR, SGPR >= 100 ? ">=100" : "few", lds, small_threads, ms, ms > 5.0 ? "BLOCKED" : "co-resident");
    hipFree(out);
    hipStreamDestroy(a);
    hipStreamDestroy(b);
}

// closer to the trace kernel: the same footprint (768 threads, 80 VGPRs, 68928 B of LDS) with, one at a time, what the spin loop lacks
template <int KIND>
__global__ void __attribute__((amdgpu_waves_per_eu(6, 6))) __launch_bounds__(1024) busy(unsigned long long ticks, float* out, unsigned* ctr) {
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v79, 0" ::: "v79");
    asm volatile("s_mov_b32 s99, 0" ::: "s99");
    if (KIND == 4) __builtin_amdgcn_s_setprio(1);
    for (unsigned i = threadIdx.x; i < 68928 / 4; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    float acc = (float)threadIdx.x;
    unsigned k = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (KIND == 1 || KIND == 5) { // VALU + LDS traffic
            for (int q = 0; q < 64; ++q) {
                k = (k * 1664525u + 1013904223u);
                acc = acc * 0.999f + lds[(k >> 8) % (68928 / 4)];
            }
        }
        if (KIND == 2 || KIND == 5) { // global stores: 16 bytes per lane, lone lines
            k = (k * 1664525u + 1013904223u);
            reinterpret_cast<float4*>(out)[(size_t)(k >> 6) % (size_t)(1u << 24)] = make_float4(acc, acc, acc, acc);
        }
        if (KIND == 3 || KIND == 5) atomicAdd(ctr, 64u); // the work counter
        acc += 1.0f;
    }
    if (acc == -1.0f) out[threadIdx.x] = acc;
}
template <int KIND>
static void run_busy(const char* what) {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    float *out, *out2;
    unsigned* ctr;
    hipMalloc(&out, (size_t)16 << 24); // 2^24 float4
    hipMalloc(&out2, 4 << 20);
    hipMalloc(&ctr, 64);
    hipMemset(ctr, 0, 64);
    const int lds = 68928;
    hipFuncSetAttribute(reinterpret_cast<const void*>(busy<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((busy<KIND>), dim3(512), dim3(768), lds, a, 2000000ull, out, ctr);
    std::this_thread::sleep_for(std::chrono::milliseconds(3));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, b, out2);
    hipStreamSynchronize(b);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(a);
    printf("busy kernel (768 threads x 2 per CU, 80 VGPRs, 68928 B LDS, waves_per_eu 6) + %-42s | small kernel done after %7.3f ms  %s\n", what, ms,
           ms > 5.0 ? "BLOCKED" : "co-resident");
    hipFree(out); hipFree(out2); hipFree(ctr);
    hipStreamDestroy(a); hipStreamDestroy(b);
}

// the library's pattern: everything queued up front -- A: [spin 10 ms][small], B: [spin 10 ms] (B's workgroups can only become
// resident as A's spin drains) -- does A's small kernel run beside B's spin or after it?
static void pipeline(int prio) {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    float* out;
    hipMalloc(&out, 4 << 20); // (small: 2048 x 256 floats = 2 MiB)
    hipEvent_t e0, e1, e2, e3;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    const int lds = 68928;
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin<80, 100>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const unsigned long long ticks = 1000000ull; // 10 ms
    hipEventRecord(e0, a);
    hipLaunchKernelGGL((spin<80, 100>), dim3(512), dim3(768), lds, a, ticks, out);
    hipEventRecord(e1, a);
    hipLaunchKernelGGL(small, dim3(2048), dim3(256), 0, a, out);
    hipEventRecord(e2, a);
    hipLaunchKernelGGL((spin<80, 100>), dim3(512), dim3(768), lds, b, ticks, out);
    hipEventRecord(e3, b);
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    float t1, t2, t3;
    hipEventElapsedTime(&t1, e0, e1); hipEventElapsedTime(&t2, e0, e2); hipEventElapsedTime(&t3, e0, e3);
    printf("pipeline: A's spin done at %6.2f ms, A's small kernel (behind it) at %6.2f ms, B's spin at %6.2f ms  -> small ran %s B's spin\n", t1, t2, t3,
           t2 < t3 - 2.0f ? "BESIDE" : "AFTER");
    (void)prio;
}

int main() {
    run_busy<0>("nothing (spin)");
    run_busy<1>("VALU + LDS traffic");
    run_busy<2>("16-byte global stores");
    run_busy<3>("atomics on one counter");
    run_busy<4>("s_setprio 1");
    run_busy<5>("all of them");
    pipeline(0);
    run<80, 100>(768, 68928, 256); // the trace kernel's footprint
    run<80, 100>(768, 68928, 64);
    run<80, 100>(768, 0, 256);
    run<16, 0>(768, 68928, 256);
    run<16, 0>(768, 0, 256);
    run<16, 100>(768, 0, 256);
    run<80, 0>(768, 0, 256);
    run<64, 0>(768, 0, 256);
    run<16, 0>(1024, 0, 256);
    run<80, 100>(704, 68928, 256);
    run<80, 100>(704, 68928, 64);
    run<80, 100>(640, 68928, 256);
    run<80, 0>(640, 0, 256);
    run<16, 0>(768, 81920, 256);
    return 0;
}
