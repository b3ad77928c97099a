// tools/probes/scratch_cost.hip -- what does a non-zero private segment cost a dispatch?  Two otherwise identical kernels,
// one with 36-48 bytes of private segment per lane that no instruction touches at run time, launched in pairs like the step's two kernels:
// a HIP graph of 2 x 500 dependent kernel nodes, timed with events.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/scratch_cost.hip -o /tmp/scratch_cost && /tmp/scratch_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool SCRATCH>
__global__ __launch_bounds__(256) void k(float *p, int n, int idx) {
    volatile float local[9];
    float x = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < n; i++) x = x * 1.0001f + 0.5f;
    if (SCRATCH && idx >= 0) {   // (idx < 0 at run time: the private segment is declared, no scratch instruction ever executes)
#pragma unroll
        for (int i = 0; i < 9; i++) local[i] = p[i];
        x += local[idx % 9];
    }
    p[blockIdx.x * 256 + threadIdx.x] = x;
}

template <bool A, bool B>
float run(float *d, int grid, int n) {
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 500; i++) {
        hipLaunchKernelGGL(k<A>, dim3(grid), dim3(256), 0, s, d, n, -1);
        hipLaunchKernelGGL(k<B>, dim3(grid), dim3(256), 0, s, d, n, -1);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1000.f / 1000.f;   // us per kernel (1000 kernels)
}

int main() {
    float *d;
    hipMalloc(&d, 1024 * 256 * 4);
    hipMemset(d, 0, 1024 * 256 * 4);
    for (int grid : {256, 768}) {
        for (int n : {100, 4000}) {
            printf("grid %4d, %4d iterations: no scratch %.2f us per kernel | one of two with 36 B scratch %.2f | both %.2f\n", grid, n,
                   run<false, false>(d, grid, n), run<true, false>(d, grid, n), run<true, true>(d, grid, n));
        }
    }
    return 0;
}
