// Calibration of rocprofv3's WRITE_SIZE for the store pattern of the sample records (guide: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel writes exactly `bytes`.
//   stream16   16 B per lane, lanes contiguous (the calibrated reference pattern)
//   pair32     each lane writes 2 x 16 B back to back into its own 32-byte sector, lanes 8 KiB apart (sample records)
//   single16   each lane writes one 16-B record, lanes 8 KiB apart (what the record store was before the pairing)
//   rec64      each lane writes 4 x 16 B = one 64-byte queue record, lanes 64 B apart (deferred-path queue)
// hipcc --offload-arch=gfx950 -O3 -o write_size_calib write_size_calib.hip ; rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void stream16(float4* p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void pair32(float4* p, size_t n_pairs, uint32_t spp) { // pixel-major records, `spp` per pixel
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_pix = n_pairs * 2 / spp;
    for (size_t k = 0; k < spp / 2; ++k) {
        if (lane < n_pix) {
            float4* dst = p + lane * spp + 2 * k;
            dst[0] = make_float4(1.f, 2.f, 3.f, (float)k);
            dst[1] = make_float4(4.f, 5.f, 6.f, (float)k);
        }
        __builtin_amdgcn_s_sleep(20); // the real kernel spends ~50 us between two pairs of one lane
    }
}
__global__ void single16(float4* p, size_t n_pix, uint32_t spp) {
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t k = 0; k < spp; ++k) {
        if (lane < n_pix) p[lane * spp + k] = make_float4(1.f, 2.f, 3.f, (float)k);
        __builtin_amdgcn_s_sleep(20);
    }
}
__global__ void rec64(float4* p, size_t n_rec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rec) {
        float4* dst = p + i * 4;
        dst[0] = make_float4(1.f, 2.f, 3.f, 4.f);
        dst[1] = make_float4(1.f, 2.f, 3.f, 5.f);
        dst[2] = make_float4(1.f, 2.f, 3.f, 6.f);
        dst[3] = make_float4(1.f, 2.f, 3.f, 7.f);
    }
}

int main() {
    const uint32_t spp = 512;
    const size_t n_pix = 1u << 18;                // 262144 pixels x 512 records x 16 B = 2 GiB
    const size_t n = n_pix * spp;
    float4* d;
    if (hipMalloc(&d, n * sizeof(float4)) != hipSuccess) return 1;
    hipMemset(d, 0, n * sizeof(float4));
    hipDeviceSynchronize();
    stream16<<<(unsigned)((n + 255) / 256), 256>>>(d, n);
    hipDeviceSynchronize();
    pair32<<<(unsigned)((n_pix + 255) / 256), 256>>>(d, n / 2, spp);
    hipDeviceSynchronize();
    single16<<<(unsigned)((n_pix + 255) / 256), 256>>>(d, n_pix, spp);
    hipDeviceSynchronize();
    rec64<<<(unsigned)((n / 4 + 255) / 256), 256>>>(d, n / 4);
    hipDeviceSynchronize();
    printf("every kernel wrote %zu bytes (%.3f GiB = %.1f KB in WRITE_SIZE units)\n", n * sizeof(float4),
           n * sizeof(float4) / 1073741824.0, n * sizeof(float4) / 1024.0);
    hipFree(d);
    return 0;
}
