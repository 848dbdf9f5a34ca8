// row_burst.hip -- what the height-map rows of an observation cost to WRITE, with no arithmetic in front of them.
// Two shapes, the two the elevation task launches:
//   burst   the fused step + scan launch at 4096 envs: 256 blocks of 512 threads, block = 16 envs, every thread 6 (5.3) 16-byte stores
//           into rows of 689 floats from column 13 -- 11.3 MB that leave all CUs at once, optionally behind a spin of S microseconds
//           (the physics the real launch waits for): per-launch time of back-to-back launches on one stream, like the bench
//   stream  the large-batch scan at 262 144 envs: block = env = 192 threads, lanes 0..168 one 16-byte store each: 709 MB
// Variants: the store's cache policy (default / nt / sc1 / sc1 nt / sc0 sc1 = write-through), plain global stores vs buffer stores,
// and the row pitch (689 floats = 2756 B, as the observation contract has it, against 704 floats = 64-byte aligned rows).
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/row_burst.hip -o <somewhere>/row_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int kQuads = 169, kEnvsPerBlock = 16;

template <int POL>
__device__ __forceinline__ void store16(float* base, __amdgpu_buffer_rsrc_t rsrc, long off_floats, f4 v) {
    if constexpr (POL == -1) {
        *reinterpret_cast<f4*>(base + off_floats) = v;
    } else if constexpr (POL == -2) {
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(base + off_floats));
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rsrc, (int)(off_floats * 4), 0, POL);
    }
}
__device__ __forceinline__ void spin_us(float us) {
    if (us <= 0.f) return;
    const unsigned long long t0 = wall_clock64();          // 100 MHz
    const unsigned long long ticks = (unsigned long long)(us * 100.f);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

template <int POL>
__global__ void __launch_bounds__(512) burst_kernel(float* __restrict__ obs, int n_envs, int pitch, int col0, float spin) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obs, 0, (int)((long)n_envs * pitch * 4), 0x00020000);
    spin_us(spin);
    const int tid = threadIdx.x, e0 = blockIdx.x * kEnvsPerBlock;
    constexpr int kAll = kEnvsPerBlock * kQuads;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = tid + k * 512;
        if (idx < kAll) {
            const int j = idx / kQuads, q = idx - j * kQuads;
            if (e0 + j < n_envs) {
                const float x = (float)idx;
                store16<POL>(obs, rsrc, (long)(e0 + j) * pitch + col0 + 4 * q, f4{x, x + 1.f, x + 2.f, x + 3.f});
            }
        }
    }
}
template <int POL>
__global__ void __launch_bounds__(192) stream_kernel(float* __restrict__ obs, int n_envs, int pitch, int col0) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obs, 0, (int)((long)n_envs * pitch * 4), 0x00020000);
    const int q = threadIdx.x, e = blockIdx.x;
    if (q < kQuads) {
        const float x = (float)q;
        store16<POL>(obs, rsrc, (long)e * pitch + col0 + 4 * q, f4{x, x + 1.f, x + 2.f, x + 3.f});
    }
}
__global__ void __launch_bounds__(512) empty_kernel(float* obs, float spin) {
    spin_us(spin);
    if (obs == nullptr) __builtin_trap();
}

static float time_launches(void (*launch)(void*), void* ctx, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) launch(ctx);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int t = 0; t < 5; ++t) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch(ctx);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best * 1000.f / (float)reps;
}
struct Ctx { float* obs; int n, pitch, col0; float spin; };
template <int POL> static void launch_burst(void* c) {
    Ctx* x = (Ctx*)c;
    burst_kernel<POL><<<(x->n + kEnvsPerBlock - 1) / kEnvsPerBlock, 512>>>(x->obs, x->n, x->pitch, x->col0, x->spin);
}
template <int POL> static void launch_stream(void* c) {
    Ctx* x = (Ctx*)c;
    stream_kernel<POL><<<x->n, 192>>>(x->obs, x->n, x->pitch, x->col0);
}
static void launch_empty(void* c) {
    Ctx* x = (Ctx*)c;
    empty_kernel<<<(x->n + kEnvsPerBlock - 1) / kEnvsPerBlock, 512>>>(x->obs, x->spin);
}

int main() {
    const int n_big = 262144;
    float* obs;
    CHECK(hipMalloc(&obs, (size_t)n_big * 704 * 4));
    CHECK(hipMemset(obs, 0, (size_t)n_big * 704 * 4));
    const char* names[] = {"global default", "global nt", "buffer default", "buffer nt", "buffer sc1", "buffer sc1 nt", "buffer sc0 sc1", "buffer sc0 sc1 nt"};
    void (*burst[])(void*) = {launch_burst<-1>, launch_burst<-2>, launch_burst<0>, launch_burst<2>, launch_burst<16>, launch_burst<18>, launch_burst<17>, launch_burst<19>};
    void (*stream[])(void*) = {launch_stream<-1>, launch_stream<-2>, launch_stream<0>, launch_stream<2>, launch_stream<16>, launch_stream<18>, launch_stream<17>, launch_stream<19>};
    for (float spin : {0.f, 12.f}) {
        Ctx c{obs, 4096, 689, 13, spin};
        const float floor_us = time_launches(launch_empty, &c, 400);
        printf("burst  n=4096 spin=%4.1f us: empty launch %.2f us\n", spin, floor_us);
        for (int pitch : {689, 704}) {
            c.pitch = pitch, c.col0 = pitch == 689 ? 13 : 16;
            for (int v = 0; v < 8; ++v) {
                const float us = time_launches(burst[v], &c, 400);
                printf("burst  n=4096 spin=%4.1f pitch=%d %-18s %.2f us per launch (+%.2f over empty; %.2f TB/s of the excess)\n", spin, pitch, names[v], us,
                       us - floor_us, 4096.0 * 676 * 4 / ((us - floor_us) * 1e6));
            }
        }
    }
    for (int n : {32768, 262144}) {
        for (int pitch : {689, 704}) {
            Ctx c{obs, n, pitch, pitch == 689 ? 13 : 16, 0.f};
            for (int v = 0; v < 8; ++v) {
                const float us = time_launches(stream[v], &c, 20);
                printf("stream n=%d pitch=%d %-18s %.1f us per launch = %.2f TB/s\n", n, pitch, names[v], us, (double)n * 676 * 4 / (us * 1e6));
            }
        }
    }
    return 0;
}
