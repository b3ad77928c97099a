// Diagnostic (GPU box; not part of the product): what the memory system gives for ROW GATHERS -- rows of D floats read at
// random / sorted / sequential row numbers out of a 3 GiB table (far beyond the 256 MiB Infinity Cache), one float4 per
// lane, UNROLL rows in flight per lane group, the rows summed into registers (reads only) or copied out (read + write).
// The planned M-step's cache-exceeding launches are priced against these rates.
// build + run (on the box): hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LG, int UNROLL, bool COPY>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ tab, const int *__restrict__ idx, float4 *__restrict__ out, int n, float *sink) {
    const int lg = threadIdx.x % LG, grp = (blockIdx.x * 256 + threadIdx.x) / LG, ngrp = gridDim.x * 256 / LG;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = grp * UNROLL; i < n; i += ngrp * UNROLL) {
        float4 r[UNROLL];
        int row[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) row[u] = i + u < n ? idx[i + u] : 0;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) r[u] = tab[(size_t)row[u] * LG + lg];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (COPY) { if (i + u < n) out[(size_t)(i + u) * LG + lg] = r[u]; }
            else { acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w; }
        }
    }
    if (!COPY && acc.x == 12345.f) *sink = acc.x + acc.y + acc.z + acc.w;
}

template <int LG, bool COPY>
void run(const char *name, const float4 *tab, const int *idx, float4 *out, int n, float *sink, int wgs) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather<LG, 4, COPY><<<wgs, 256>>>(tab, idx, out, n, sink);
    hipEventRecord(a);
    for (int k = 0; k < 5; k++) gather<LG, 4, COPY><<<wgs, 256>>>(tab, idx, out, n, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double bytes = (double)n * LG * 16 * (COPY ? 2 : 1);
    printf("D=%3d %-10s %-5s %d workgroups: %7.3f ms per %d rows -> %5.2f TB/s\n", LG * 4, name, COPY ? "copy" : "read", wgs, ms, n, bytes / ms / 1e9);
}

template <int LG>
void sweep() {
    const size_t rows = ((size_t)3 << 30) / (16 * LG);
    const int n = (int)(((size_t)1 << 30) / (16 * LG));
    float4 *tab, *out;
    int *idx;
    float *sink;
    hipMalloc(&tab, rows * LG * 16); hipMalloc(&out, (size_t)n * LG * 16); hipMalloc(&idx, n * 4); hipMalloc(&sink, 4);
    hipMemset(tab, 0, rows * LG * 16);
    std::vector<int> h(n);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = (int)((((size_t)rand() << 16) ^ rand()) % rows);
    for (int pass = 0; pass < 3; pass++) {
        if (pass == 1) std::sort(h.begin(), h.end());
        if (pass == 2) for (int i = 0; i < n; i++) h[i] = i + 4321;
        hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
        const char *name = pass == 0 ? "random" : (pass == 1 ? "sorted" : "sequential");
        for (int wgs : {2048, 8192}) {
            run<LG, false>(name, tab, idx, out, n, sink, wgs);
            run<LG, true>(name, tab, idx, out, n, sink, wgs);
        }
    }
    hipFree(tab); hipFree(out); hipFree(idx); hipFree(sink);
}

int main() {
    sweep<16>();
    sweep<32>();
    sweep<64>();
    return 0;
}
