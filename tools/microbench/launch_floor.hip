// Microbenchmark: what does ONE dependent launch cost at the bench workload's shape (4096 envs in quad form = 64 blocks
// of 256 threads) before any arithmetic?  (A) empty kernel, (B) 34 row loads + 1 store, (C) 34 loads + 30 row stores,
// (D) = C plus a dependent chain of N fma (stand-in for the physics) -- back to back in one stream, us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_empty(float* base, long stride, int n, int chain) {}

template <int R, int W>
__global__ void __launch_bounds__(256) k_rows(float* __restrict__ base, long stride, int n, int chain) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;     // quad form: 4 lanes per env
    if (e >= n) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * 41), 0x00020000);
    const int voff = e * 4, rowb = (int)(stride * 4);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, 0));
    for (int i = 0; i < chain; ++i) s = fmaf(s, 1.0000001f, 1e-9f);      // dependent chain: ~ chain * issue latency
    if ((threadIdx.x & 3) == 0) {
#pragma unroll
        for (int w = 0; w < W; ++w) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s + (float)w), rs, voff, w * rowb, 0);
    }
}

// (F) kernel-argument fetch: a 640-byte by-value struct whose every dword is consumed (the step kernel's kernarg
// segment is ~650 B) vs (C)'s 24 bytes
struct BigArgs { float f[160]; };
__global__ void __launch_bounds__(256) k_bigargs(const BigArgs a, float* __restrict__ base, long stride, int n) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;
    if (e >= n) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 160; ++i) s += a.f[i];
    if ((threadIdx.x & 3) == 0) base[e] = s;
}

// (G) as F but the pointers come FIRST (preloadable into SGPRs with -mllvm -amdgpu-kernarg-preload-count=N) and the
// kernel does its 34 row loads before it needs the struct: can the struct's fetch overlap the row loads?
__global__ void __launch_bounds__(256) k_bigargs_ptr_first(float* __restrict__ base, long stride, int n, const BigArgs a) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;
    if (e >= n) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * 41), 0x00020000);
    const int voff = e * 4, rowb = (int)(stride * 4);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 34; ++r) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, 0));
#pragma unroll
    for (int i = 0; i < 160; ++i) s += a.f[i];
    if ((threadIdx.x & 3) == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rs, voff, 0, 0);
}

// (H) as F but the 640 bytes live in an ordinary (cached) device buffer and are read with scalar loads through a
// constant-address-space pointer: kernarg memory is written by the host for every launch and read uncached
__global__ void __launch_bounds__(256) k_devargs(const BigArgs* __restrict__ a_dev, float* __restrict__ base, long stride, int n) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;
    if (e >= n) return;
    const __attribute__((address_space(4))) float* f = (const __attribute__((address_space(4))) float*)a_dev;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 160; ++i) s += f[i];
    if ((threadIdx.x & 3) == 0) base[e] = s;
}

// (I) the candidate design: parameters in a device buffer, fetched with VECTOR loads at a lane-uniform address (into
// VGPRs, where float parameters are consumed anyway) in the same batch as the 34 state-row loads
__global__ void __launch_bounds__(256) k_vecargs(const float* __restrict__ a_dev, float* __restrict__ base, long stride, int n) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;
    if (e >= n) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * 41), 0x00020000);
    const int voff = e * 4, rowb = (int)(stride * 4);
    int z = 0;
    asm volatile("" : "+v"(z));          // a VGPR zero the compiler cannot see through: keeps the loads on the vector path
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 34; ++r) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, 0));
#pragma unroll
    for (int i = 0; i < 160; ++i) s += a_dev[i + z];
    if ((threadIdx.x & 3) == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rs, voff, 0, 0);
}

// (J) as I but the 640 bytes are the kernel's own by-value argument, read with vector loads from the kernarg segment
__global__ void __launch_bounds__(256) k_vec_kernarg(const BigArgs a, float* __restrict__ base, long stride, int n) {
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 2;
    if (e >= n) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * 41), 0x00020000);
    const int voff = e * 4, rowb = (int)(stride * 4);
    const __attribute__((address_space(4))) float* ka = (const __attribute__((address_space(4))) float*)__builtin_amdgcn_kernarg_segment_ptr();
    int z = 0;
    asm volatile("" : "+v"(z));
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 34; ++r) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, r * rowb, 0));
#pragma unroll
    for (int i = 0; i < 160; ++i) s += ka[i + z];
    if ((threadIdx.x & 3) == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rs, voff, 0, 0);
}

template <class K>
float run(K kern, float* buf, long stride, int n, int chain, int grid) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) kern<<<grid, 256>>>(buf, stride, n, chain);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 1000; ++i) kern<<<grid, 256>>>(buf, stride, n, chain);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;   // 1000 launches: ms == us per launch
}

int main() {
    const int n = 4096;
    const long stride = 4096;
    float* buf;
    CHECK(hipMalloc(&buf, stride * 4 * 41));
    CHECK(hipMemset(buf, 0, stride * 4 * 41));
    const int grid = n * 4 / 256;
    printf("A empty                       : %.2f us/launch\n", run(k_empty, buf, stride, n, 0, grid));
    printf("B 34 loads + 1 store          : %.2f\n", run(k_rows<34, 1>, buf, stride, n, 0, grid));
    printf("C 34 loads + 30 stores        : %.2f\n", run(k_rows<34, 30>, buf, stride, n, 0, grid));
    for (int chain : {250, 500, 1000, 2000, 4000})
        printf("D C + %4d dependent fma      : %.2f\n", chain, run(k_rows<34, 30>, buf, stride, n, chain, grid));
    printf("E empty, 1 block              : %.2f\n", run(k_empty, buf, stride, n, 0, 1));
    {
        BigArgs a;
        for (int i = 0; i < 160; ++i) a.f[i] = (float)i;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) k_bigargs<<<grid, 256>>>(a, buf, stride, n);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) k_bigargs<<<grid, 256>>>(a, buf, stride, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("F 640-byte kernarg, all read  : %.2f\n", ms);
        for (int i = 0; i < 20; ++i) k_bigargs_ptr_first<<<grid, 256>>>(buf, stride, n, a);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) k_bigargs_ptr_first<<<grid, 256>>>(buf, stride, n, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        BigArgs* a_dev;
        CHECK(hipMalloc(&a_dev, sizeof(BigArgs)));
        CHECK(hipMemcpy(a_dev, &a, sizeof(BigArgs), hipMemcpyHostToDevice));
        for (int i = 0; i < 20; ++i) k_devargs<<<grid, 256>>>(a_dev, buf, stride, n);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) k_devargs<<<grid, 256>>>(a_dev, buf, stride, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        { float ms2; CHECK(hipEventElapsedTime(&ms2, e0, e1)); printf("H 640 bytes from a device buffer (s_load) : %.2f\n", ms2); }
        for (int i = 0; i < 20; ++i) k_vecargs<<<grid, 256>>>((const float*)a_dev, buf, stride, n);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) k_vecargs<<<grid, 256>>>((const float*)a_dev, buf, stride, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        { float ms2; CHECK(hipEventElapsedTime(&ms2, e0, e1)); printf("I 34 row loads + 640 bytes by vector loads from a device buffer : %.2f\n", ms2); }
        for (int i = 0; i < 20; ++i) k_vec_kernarg<<<grid, 256>>>(a, buf, stride, n);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) k_vec_kernarg<<<grid, 256>>>(a, buf, stride, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        { float ms2; CHECK(hipEventElapsedTime(&ms2, e0, e1)); printf("J 34 row loads + 640 bytes by vector loads from the kernarg segment : %.2f\n", ms2); }
        printf("G pointers first + 34 loads + 640-byte struct : %.2f\n", ms);
    }
    return 0;
}
