// Microbenchmark: what one VALU instruction costs a wavefront that is ALONE on its SIMD (the quad-form kernels at
// 4096 envs): dependent chain vs 8 independent chains, scalar fp32 fma vs packed v_pk_fma_f32, v_rcp_f32, DPP add.
// 64 blocks x 256 threads (one wave per SIMD on 64 CUs), 2^20 instructions per wave, timed with events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kUnroll = 256, kIters = 4096;   // 2^20 instructions

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x[8];
    f2 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = a + threadIdx.x * 1e-6f + j; p[j] = f2{x[j], x[j] + 1.f}; }
    const f2 a2{a, a}, b2{b, b};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));           // dependent
            if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u & 7]) : "v"(a), "v"(b));       // 8 chains
            if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(a2), "v"(b2));
            if constexpr (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u & 7]) : "v"(a2), "v"(b2));
            if constexpr (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[0]));
            if constexpr (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[u & 7]));
            if constexpr (MODE == 6) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[0]));
            if constexpr (MODE == 7) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[u & 7]));
            if constexpr (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(x[u & 7]) : "v"(x[(u + 1) & 7]));
            if constexpr (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u & 7]) : "v"(a) : );
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// waves_per_simd = 1: 256 blocks (one per CU, one wave per SIMD); 4: 1024 blocks (four waves per SIMD).  Reported: SIMD time
// per instruction = kernel time / (instructions per wave x waves per SIMD)
template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double n = (double)kUnroll * kIters;
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        k<MODE><<<grid, 256>>>(out, 0.999f, 1e-3f);
        CHECK(hipEventRecord(e0));
        k<MODE><<<grid, 256>>>(out, 0.999f, 1e-3f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d wave/SIMD: %.3f ns", wps, ms * 1e6 / (n * wps));
    }
    printf("   (SIMD time per instruction)\n");
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 1024 * 256 * 4));
    run<0>("v_fma_f32, dependent chain", out);
    run<1>("v_fma_f32, 8 independent chains", out);
    run<2>("v_pk_fma_f32, dependent chain", out);
    run<3>("v_pk_fma_f32, 8 independent chains", out);
    run<4>("v_rcp_f32, dependent chain", out);
    run<5>("v_rcp_f32, 8 independent chains", out);
    run<6>("v_add_f32_dpp, dependent chain", out);
    run<7>("v_add_f32_dpp, 8 independent", out);
    run<8>("v_mov_b32, independent", out);
    run<9>("v_cndmask_b32, independent", out);
    return 0;
}
