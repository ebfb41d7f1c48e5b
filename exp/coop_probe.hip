// Probe: does a cooperative launch with grid-wide barriers work on this box (and under rocprofv3)?
//   hipcc --offload-arch=gfx950 -O2 exp/coop_probe.hip -o exp/bin/coop_probe && timeout -k 5 60 exp/bin/coop_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256, 3) probe(unsigned *ctr, unsigned *out, int rounds) {
    cg::grid_group grid = cg::this_grid();
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) atomicAdd(ctr + r, 1u);
        grid.sync();
        if (blockIdx.x == 0 && threadIdx.x == 0) out[r] = ctr[r];     // every workgroup has arrived
        grid.sync();
    }
}

int main() {
    int dev = 0, coop = 0, cus = 0, per_cu = 0;
    hipSetDevice(dev);
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, probe, 256, 0);
    printf("cooperative launch attribute %d, CUs %d, blocks per CU %d\n", coop, cus, per_cu);
    const int rounds = 50, grid = cus * per_cu;
    unsigned *ctr, *out;
    hipMalloc(&ctr, rounds * 4); hipMalloc(&out, rounds * 4);
    hipMemset(ctr, 0, rounds * 4); hipMemset(out, 0, rounds * 4);
    int r = rounds;
    void *args[] = {&ctr, &out, &r};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipError_t e = hipLaunchCooperativeKernel((const void *)probe, dim3(grid), dim3(256), args, 0, 0);
    hipEventRecord(e1, 0);
    printf("launch: %s\n", hipGetErrorString(e));
    e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[50]; hipMemcpy(h, out, rounds * 4, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < rounds; ++i) ok = ok && (h[i] == (unsigned)grid);
    printf("grid %d, %d rounds of 2 barriers: %s, %.3f ms (%.2f us per barrier)\n", grid, rounds, ok ? "OK" : "WRONG", ms, ms * 1e3 / (2 * rounds));
    // an oversubscribed cooperative launch must be refused, not hang
    e = hipLaunchCooperativeKernel((const void *)probe, dim3(grid * 4), dim3(256), args, 0, 0);
    printf("oversubscribed launch: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return ok ? 0 : 1;
}
