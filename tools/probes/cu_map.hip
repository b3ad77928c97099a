// tools/probes/cu_map.hip -- which workgroups of a launch share a CU?  Every workgroup records the XCC / SE / SH / CU it runs
// on (HW_REG_XCC_ID, HW_REG_HW_ID) and spins long enough for the whole grid to be resident at once.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_map.hip -o /tmp/cu_map && /tmp/cu_map [grid] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, long long spin) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    const int lds = argc > 2 ? atoi(argv[2]) : 70 * 1024;
    unsigned *d;
    hipMalloc(&d, grid * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 3000LL);   // 100 MHz wall clock: 30 us
    std::vector<unsigned> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < grid; b++) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        if (b < 24) printf("wg %3d: xcc %u se %u sh %u cu %2u\n", b, xcc, se, sh, cu);
    }
    printf("%zu distinct CUs\n", by_cu.size());
    std::map<int, int> delta;
    for (auto &kv : by_cu)
        for (size_t i = 1; i < kv.second.size(); i++) delta[kv.second[i] - kv.second[i - 1]]++;
    for (auto &kv : delta) printf("co-resident workgroups %d apart: %d pairs\n", kv.first, kv.second);
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 6) break;
        printf("cu %04x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    return 0;
}
