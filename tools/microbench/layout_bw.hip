// Microbenchmark: streaming R rows in / W rows out per env in (a) SoA [row][n] and (b) AoSoA [n/64][row][64] layouts.
// Answers: is the fused step's memory floor (4.2 TB/s at 4 M envs) a property of the many-stream SoA pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// AUXL / AUXS: the cache-policy operand of the buffer loads / stores (gfx940+: bit 0 sc0, bit 1 nt, bit 4 sc1)
template <int R, int W, bool TILED, int AUXL = 0, int AUXS = 0>
__global__ void __launch_bounds__(256) stream_kernel(float* __restrict__ base, long stride, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    constexpr int ROWS = 41;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * ROWS), 0x00020000);
    const int voff = TILED ? ((e >> 6) * (ROWS * 256) + (e & 63) * 4) : e * 4;
    const int rowb = TILED ? 256 : (int)(stride * 4);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, AUXL));
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
#pragma unroll
    for (int w = 0; w < W; ++w) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s + (float)w), rs, voff, w * rowb, AUXS);
}

// (c) rows grouped in fours: [group][n] of float4 -- 16 B per lane per access, 1 KB contiguous per wavefront
typedef float f4 __attribute__((ext_vector_type(4)));
template <int GR, int GW>
__global__ void __launch_bounds__(256) stream4_kernel(f4* __restrict__ base, long stride, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    f4 acc[GR];
#pragma unroll
    for (int g = 0; g < GR; ++g) acc[g] = base[(long)g * stride + e];
    f4 s = acc[0];
#pragma unroll
    for (int g = 1; g < GR; ++g) s += acc[g];
#pragma unroll
    for (int g = 0; g < GW; ++g) base[(long)g * stride + e] = s + (float)g;
}
template <int GR, int GW>
float run4(float* buf, long stride, int n) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) stream4_kernel<GR, GW><<<(n + 255) / 256, 256>>>((f4*)buf, stride, n);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) stream4_kernel<GR, GW><<<(n + 255) / 256, 256>>>((f4*)buf, stride, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 100.f;
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

template <int R, int W>
float run_rw(float* buf, long stride, int n) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) stream_kernel<R, W, false><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) stream_kernel<R, W, false><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 100.f;
}
__global__ void __launch_bounds__(256) copy4_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) copy4_nt_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
template <int AUXL, int AUXS>
float run_policy(float* buf, long stride, int n) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) stream_kernel<34, 30, false, AUXL, AUXS><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) stream_kernel<34, 30, false, AUXL, AUXS><<<(n + 255) / 256, 256>>>(buf, stride, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 100.f;
}

int main() {
    {   // round 3: (a) the guide's float4 copy at a size well past the 256 MB Infinity Cache, plain and non-temporal; (b) the cache
        // policy bits on the step kernel's own pattern (34 SoA rows in, 30 out, 4 B per lane, 4 M envs)
        const long n4 = (1L << 30) / 16;     // 1 GiB -> 1 GiB
        f4 *src, *dst;
        CHECK(hipMalloc(&src, n4 * 16)); CHECK(hipMalloc(&dst, n4 * 16));
        CHECK(hipMemset(src, 0, n4 * 16)); CHECK(hipMemset(dst, 0, n4 * 16));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int nt = 0; nt < 2; ++nt)
            for (int g : {2048, 8192, 65536}) {
                for (int i = 0; i < 2; ++i) nt ? copy4_nt_kernel<<<g, 256>>>(src, dst, n4) : copy4_kernel<<<g, 256>>>(src, dst, n4);
                CHECK(hipEventRecord(a));
                for (int i = 0; i < 5; ++i) nt ? copy4_nt_kernel<<<g, 256>>>(src, dst, n4) : copy4_kernel<<<g, 256>>>(src, dst, n4);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                printf("float4 copy 1 GiB -> 1 GiB%s, grid %d: %.1f us (%.2f TB/s)\n", nt ? " non-temporal" : "", g, ms * 200.f,
                       2.0 * n4 * 16 / (ms * 200.f) / 1e6);
            }
        CHECK(hipFree(src)); CHECK(hipFree(dst));
        const int n = 1 << 22;
        float* buf;
        CHECK(hipMalloc(&buf, (long)n * 4 * 44));
        CHECK(hipMemset(buf, 0, (long)n * 4 * 44));
        const double bytes = 64.0 * 4 * n;
        const float p00 = run_policy<0, 0>(buf, n, n), p02 = run_policy<0, 2>(buf, n, n), p22 = run_policy<2, 2>(buf, n, n),
                    p20 = run_policy<2, 0>(buf, n, n), p016 = run_policy<0, 16>(buf, n, n), p018 = run_policy<0, 18>(buf, n, n), p218 = run_policy<2, 18>(buf, n, n),
                    p1818 = run_policy<18, 18>(buf, n, n);
        printf("SoA 34 in / 30 out, 4 M envs, cache policy (loads, stores): default %.1f us (%.2f TB/s)  (-, nt) %.1f (%.2f)  (nt, nt) %.1f (%.2f)  "
               "(nt, -) %.1f (%.2f)  (-, sc1) %.1f (%.2f)  (-, sc1 nt) %.1f (%.2f)  (nt, sc1 nt) %.1f (%.2f)  (sc1 nt, sc1 nt) %.1f (%.2f)\n", p00, bytes / p00 / 1e6, p02, bytes / p02 / 1e6, p22,
               bytes / p22 / 1e6, p20, bytes / p20 / 1e6, p016, bytes / p016 / 1e6, p018, bytes / p018 / 1e6, p218, bytes / p218 / 1e6, p1818,
               bytes / p1818 / 1e6);
        CHECK(hipFree(buf));
    }
    {   // reference points on this box: plain float4 copy (grid-stride, 16 B / lane), read-mostly and write-mostly SoA streams
        const int n = 1 << 22;
        float* buf;
        CHECK(hipMalloc(&buf, (long)n * 4 * 44));
        CHECK(hipMemset(buf, 0, (long)n * 4 * 44));
        const long n4 = (long)n * 20 / 4;   // 20 rows -> 20 rows
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int g : {4096, 16384, 65536}) {
            for (int i = 0; i < 3; ++i) copy4_kernel<<<g, 256>>>((const f4*)buf, (f4*)buf + n4, n4);
            CHECK(hipEventRecord(a));
            for (int i = 0; i < 10; ++i) copy4_kernel<<<g, 256>>>((const f4*)buf, (f4*)buf + n4, n4);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            printf("float4 copy 336 MB -> 336 MB, grid %d: %.1f us (%.2f TB/s)\n", g, ms * 100.f, 2.0 * n4 * 16 / (ms * 100.f) / 1e6);
        }
        const float r = run_rw<34, 1>(buf, n, n), w = run_rw<1, 30>(buf, n, n), h = run_rw<17, 15>(buf, n, n);
        printf("SoA 34 in / 1 out %.1f us (%.2f TB/s)   1 in / 30 out %.1f us (%.2f TB/s)   17 in / 15 out %.1f us (%.2f TB/s)\n", r,
               35.0 * 4 * n / r / 1e6, w, 31.0 * 4 * n / w / 1e6, h, 32.0 * 4 * n / h / 1e6);
        CHECK(hipFree(buf));
    }
    for (int n : {1 << 20, 1 << 22}) {
        const long stride = n;
        float* buf;
        CHECK(hipMalloc(&buf, stride * 4 * 44));
        CHECK(hipMemset(buf, 0, stride * 4 * 44));
        const double bytes = (34.0 + 30.0) * 4 * n;
        const float us_soa = run<false>(buf, stride, n), us_tiled = run<true>(buf, stride, n);
        const float us_g4 = run4<9, 8>(buf, stride, n);
        const double bytes4 = (9.0 + 8.0) * 16 * n;
        printf("n=%d  SoA %.1f us (%.2f TB/s)   AoSoA64 %.1f us (%.2f TB/s)   float4 groups 9 in / 8 out %.1f us (%.2f TB/s)\n", n,
               us_soa, bytes / us_soa / 1e6, us_tiled, bytes / us_tiled / 1e6, us_g4, bytes4 / us_g4 / 1e6);
        CHECK(hipFree(buf));
    }
    return 0;
}
