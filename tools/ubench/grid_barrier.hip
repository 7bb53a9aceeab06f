// grid_barrier.hip -- what a software grid barrier costs on MI355X (why the single-pair path was not turned into
// persistent kernels).  N resident workgroups run K rounds; in a round every workgroup stores a word that ANOTHER workgroup (usually on
// another XCD: ids are dealt round robin over the 8 XCDs) loads after the barrier and checks.  The barrier: __syncthreads, thread 0
// does an agent-scope release fence, an atomic increment of one counter in device memory, spins (bounded: never hangs) until all N have
// arrived, an agent-scope acquire fence, __syncthreads.  Prints microseconds per round for N = 64 .. 1024 and the number of stale loads.
// build: hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip      run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_rounds(unsigned* cnt, unsigned* buf, unsigned* stale, unsigned* timeout, int rounds, int fence)
{
    const unsigned n = gridDim.x, me = blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        if (threadIdx.x == 0) buf[(r & 1) * 2048 + me] = (unsigned)r * 4096u + me;      // plain store, as the NNF / cost planes would be (two halves:
                                                                                       // a neighbour one barrier ahead writes the other one)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(cnt, 1u);
            const unsigned target = (unsigned)(r + 1) * n;
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { atomicAdd(timeout, 1u); break; }    // bounded: a partially resident grid must not hang the box
            }
            if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned other = (me + 1) % n;                              // the neighbour id sits on the next XCD
            if (buf[(r & 1) * 2048 + other] != (unsigned)r * 4096u + other) atomicAdd(stale, 1u);
        }
        __syncthreads();
    }
}

int main()
{
    unsigned *cnt, *buf, *stale, *timeout;
    hipMalloc(&cnt, 4); hipMalloc(&buf, 4096 * 4); hipMalloc(&stale, 4); hipMalloc(&timeout, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 200;
    for (int fence = 1; fence >= 0; fence--)
        for (int n : {64, 128, 256, 512, 1024}) {
            float best = 1e30f;
            unsigned hs = 0, ht = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(cnt, 0, 4); hipMemset(stale, 0, 4); hipMemset(timeout, 0, 4); hipMemset(buf, 0xff, 4096 * 4);
                hipEventRecord(a);
                hipLaunchKernelGGL(k_rounds, dim3(n), dim3(256), 0, 0, cnt, buf, stale, timeout, rounds, fence);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
                unsigned s = 0, t = 0;
                hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost); hipMemcpy(&t, timeout, 4, hipMemcpyDeviceToHost);
                hs += s; ht += t;
            }
            printf("%s  %4d workgroups: %6.2f us per round (store + barrier + neighbour load), stale loads %u, timeouts %u\n",
                   fence ? "agent-scope fences" : "NO fences (timing)", n, best * 1e3f / rounds, hs, ht);
        }
    return 0;
}
