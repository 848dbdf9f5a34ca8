// pmc_calib.hip -- known HBM bytes for the calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE counters (tools/profile_pmc.sh,
// tools/pmc_report.py): two kernels whose traffic is exactly what their names say, far beyond every cache (256 MB Infinity Cache).
//   calib_copy4_kernel      1 GiB -> 1 GiB, one 16-byte load and store per thread                 (read 2^30 B, write 2^30 B)
//   calib_soa_dword_kernel  the drift step's own access shape: 34 SoA rows in, 30 rows out, one dword per lane and row,
//                           4 194 304 columns                                     (read 34 x 4 x 2^22 B, write 30 x 4 x 2^22 B)
// Each runs 2 warm-up + 5 measured launches; pmc_report.py drops the first dispatches and averages the rest.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/pmc_calib.hip -o <somewhere>/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr long kCopyBytes = 1L << 30;
constexpr int kCols = 1 << 22, kRowsIn = 34, kRowsOut = 30;

__global__ void __launch_bounds__(256) calib_copy4_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) calib_soa_dword_kernel(const float* __restrict__ in, float* __restrict__ out, long stride) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < kRowsIn; ++r) acc += in[r * stride + e];
#pragma unroll
    for (int r = 0; r < kRowsOut; ++r) out[r * stride + e] = acc + (float)r;
}

int main() {
    f4 *src, *dst;
    CHECK(hipMalloc(&src, kCopyBytes));
    CHECK(hipMalloc(&dst, kCopyBytes));
    CHECK(hipMemset(src, 0, kCopyBytes));
    const long n4 = kCopyBytes / 16;
    for (int i = 0; i < 7; ++i) calib_copy4_kernel<<<(unsigned)(n4 / 256), 256>>>(src, dst, n4);
    CHECK(hipDeviceSynchronize());
    float* in = (float*)src;      // 34 x 16 MiB = 544 MiB of the 1 GiB buffers
    float* out = (float*)dst;
    for (int i = 0; i < 7; ++i) calib_soa_dword_kernel<<<kCols / 256, 256>>>(in, out, kCols);
    CHECK(hipDeviceSynchronize());
    printf("pmc_calib: copy4 read %ld write %ld ; soa_dword read %ld write %ld bytes per launch\n", kCopyBytes, kCopyBytes,
           (long)kRowsIn * 4 * kCols, (long)kRowsOut * 4 * kCols);
    return 0;
}
