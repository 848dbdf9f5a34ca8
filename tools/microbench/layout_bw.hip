// Microbenchmark: streaming R rows in / W rows out per env in (a) SoA [row][n] and (b) AoSoA [n/64][row][64] layouts.
// Answers: is the fused step's memory floor (4.2 TB/s at 4 M envs) a property of the many-stream SoA pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int R, int W, bool TILED>
__global__ void __launch_bounds__(256) stream_kernel(float* __restrict__ base, long stride, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    constexpr int ROWS = 41;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * ROWS), 0x00020000);
    const int voff = TILED ? ((e >> 6) * (ROWS * 256) + (e & 63) * 4) : e * 4;
    const int rowb = TILED ? 256 : (int)(stride * 4);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, 0));
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
#pragma unroll
    for (int w = 0; w < W; ++w) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s + (float)w), rs, voff, w * rowb, 0);
}

template <bool TILED>
float run(float* buf, long stride, int n) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) stream_kernel<34, 30, TILED><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) stream_kernel<34, 30, TILED><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 100.f;   // us per launch
}

int main() {
    for (int n : {1 << 20, 1 << 22}) {
        const long stride = n;
        float* buf;
        CHECK(hipMalloc(&buf, stride * 4 * 41));
        CHECK(hipMemset(buf, 0, stride * 4 * 41));
        const double bytes = (34.0 + 30.0) * 4 * n;
        const float us_soa = run<false>(buf, stride, n), us_tiled = run<true>(buf, stride, n);
        printf("n=%d  SoA %.1f us (%.2f TB/s)   AoSoA64 %.1f us (%.2f TB/s)\n", n, us_soa, bytes / us_soa / 1e6, us_tiled,
               bytes / us_tiled / 1e6);
        CHECK(hipFree(buf));
    }
    return 0;
}
