// launch_chain.hip -- can consecutive env.step() launches overlap?  Block b of step k + 1 depends only on block b of step k
// (the same 64 envs).  (A) the usual way: one stream, each launch waits for the previous one to drain.  (B) alternate
// two streams and chain the launches PER BLOCK through an agent-scope release / acquire flag, so that the dispatch and
// prologue of step k + 1 run while step k is still integrating.  The kernel mimics the shape of the fused drift step at
// 4096 envs (64 blocks x 256 lanes, 34 row loads, a dependent FMA chain of ~4.5 us, 34 row stores) and evolves its
// rows deterministically so that a stale read shows up as a wrong final value.  Every spin is bounded.
// build: hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 34, kBlocks = 64, kThreads = 256, kN = kBlocks * kThreads;

__global__ void __launch_bounds__(kThreads) step_kernel(float* __restrict__ rows, unsigned* flags, unsigned* err, unsigned step,
                                                        int chained, int chain_len) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (chained && step > 0) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[blockIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < step) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 2000000u) {   // ~ tens of ms: give up loudly instead of hanging the GPU
                    atomicAdd(err, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
    float v[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) v[r] = rows[r * kN + i];
    float acc = v[0];
    for (int k = 0; k < chain_len; ++k) acc = fmaf(acc, 1.0000001f, 1e-7f);   // a dependent chain: ~8 cycles per step when alone
    const float bump = acc - acc + 1.f;                                      // == 1 (keeps the chain alive)
#pragma unroll
    for (int r = 0; r < kRows; ++r) rows[r * kN + i] = v[r] + bump * (float)(r + 1);
    if (chained) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_store(&flags[blockIdx.x], step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512, chain_len = argc > 2 ? atoi(argv[2]) : 1400;
    float* rows;
    unsigned *flags, *err;
    CK(hipMalloc(&rows, sizeof(float) * kRows * kN));
    CK(hipMalloc(&flags, sizeof(unsigned) * kBlocks));
    CK(hipMalloc(&err, sizeof(unsigned)));
    hipStream_t s[2];
    CK(hipStreamCreate(&s[0]));
    CK(hipStreamCreate(&s[1]));
    hipEvent_t e0, e1, fork, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
    // (C) the serial chain captured once into a hipGraph and replayed: does the graph executor close the gap between
    // dependent kernel nodes?
    hipGraph_t graph;
    hipGraphExec_t gexec;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) step_kernel<<<kBlocks, kThreads, 0, s[0]>>>(rows, flags, err, (unsigned)k, 0, chain_len);
    CK(hipStreamEndCapture(s[0], &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(rows, 0, sizeof(float) * kRows * kN, s[0]));
            CK(hipMemsetAsync(flags, 0, sizeof(unsigned) * kBlocks, s[0]));
            CK(hipMemsetAsync(err, 0, sizeof(unsigned), s[0]));
            CK(hipEventRecord(e0, s[0]));
            if (mode == 0) {
                for (int k = 0; k < K; ++k) step_kernel<<<kBlocks, kThreads, 0, s[0]>>>(rows, flags, err, (unsigned)k, 0, chain_len);
            } else if (mode == 2) {
                CK(hipGraphLaunch(gexec, s[0]));
            } else {
                CK(hipEventRecord(fork, s[0]));
                CK(hipStreamWaitEvent(s[1], fork, 0));
                for (int k = 0; k < K; ++k) step_kernel<<<kBlocks, kThreads, 0, s[k & 1]>>>(rows, flags, err, (unsigned)k, 1, chain_len);
                CK(hipEventRecord(join, s[1]));
                CK(hipStreamWaitEvent(s[0], join, 0));
            }
            CK(hipEventRecord(e1, s[0]));
            CK(hipStreamSynchronize(s[0]));
            CK(hipStreamSynchronize(s[1]));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<float> h(kRows * kN);
            unsigned herr = 0;
            CK(hipMemcpy(h.data(), rows, sizeof(float) * kRows * kN, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&herr, err, sizeof(unsigned), hipMemcpyDeviceToHost));
            long bad = 0;
            for (int r = 0; r < kRows; ++r)
                for (int i = 0; i < kN; ++i) bad += h[r * kN + i] != (float)(K * (r + 1));
            printf("%s rep %d: %.2f us per step, %ld wrong values, %u spin time-outs\n",
                   mode == 0 ? "serial 1-stream" : mode == 1 ? "chained 2-stream" : "hipGraph replay ", rep,
                   ms * 1e3 / K, bad, herr);
        }
    }
    return 0;
}
