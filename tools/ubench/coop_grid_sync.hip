// coop_grid_sync.hip -- what cooperative groups' grid.sync() costs on MI355X (hipLaunchCooperativeKernel: the runtime checks that the
// grid fits and runs it on the device's cooperative queue), next to the software barrier of grid_barrier.hip.  N workgroups run K
// rounds; in a round every workgroup stores a word that ANOTHER workgroup (usually on another XCD) loads after the sync and checks.
// Prints microseconds per round and the number of stale loads, alone and with a long VALU kernel running on another stream.
// build: hipcc --offload-arch=gfx950 -O2 -o coop_grid_sync coop_grid_sync.hip      run: ./coop_grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>

namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_rounds(unsigned* buf, unsigned* stale, int rounds)
{
    cg::grid_group grid = cg::this_grid();
    const unsigned n = gridDim.x, me = blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        if (threadIdx.x == 0) buf[(r & 1) * 2048 + me] = (unsigned)r * 4096u + me;
        grid.sync();
        if (threadIdx.x == 0) {
            const unsigned other = (me + 1) % n;
            if (buf[(r & 1) * 2048 + other] != (unsigned)r * 4096u + other) atomicAdd(stale, 1u);
        }
    }
}
__global__ void k_busy(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) a = a * b + 0.5f;
    if (a == 123.0f) out[0] = a;
}

int main()
{
    unsigned *buf, *stale;
    float* sink;
    hipMalloc(&buf, 4096 * 4); hipMalloc(&stale, 4); hipMalloc(&sink, 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int rounds = 200;
    for (int busy = 0; busy < 2; busy++)
        for (int n : {16, 64, 256, 512, 1024}) {
            float best = 1e30f;
            unsigned hs = 0;
            hipError_t err = hipSuccess;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(stale, 0, 4); hipMemset(buf, 0xff, 4096 * 4);
                hipDeviceSynchronize();
                if (busy) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, sink, 400000);      // ~ms of VALU work on every CU
                void* args[] = {&buf, &stale, &rounds};
                hipEventRecord(a, s1);
                err = hipLaunchCooperativeKernel((const void*)k_rounds, dim3(n), dim3(256), args, 0, s1);
                hipEventRecord(b, s1);
                if (err != hipSuccess) break;
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
                unsigned s = 0;
                hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost);
                hs += s;
                hipDeviceSynchronize();
            }
            if (err != hipSuccess) { printf("%4d workgroups: launch refused: %s\n", n, hipGetErrorString(err)); (void)hipGetLastError(); continue; }
            printf("%s %4d workgroups: %6.2f us per round (store + grid.sync + neighbour load), stale loads %u\n",
                   busy ? "beside a VALU kernel on another stream," : "alone,                                ", n, best * 1e3f / rounds, hs);
        }
    return 0;
}
