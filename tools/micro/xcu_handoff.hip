// Developer micro-benchmark (not part of the library): what does ONE cross-workgroup exchange per FPS round cost on an
// otherwise idle MI355X?  G workgroups on different CUs act as the partners that would share one cloud's buckets
// (SURVEY.md §7.3 option a).  Per round every partner publishes a candidate with a 64-bit atomicMax into the group's
// slot (keys carry the round number, so the slot never has to be reset), bumps the group's arrival counter with release
// semantics, spins until all G have arrived and reads the winner back — the minimum a round of "several workgroups per
// cloud" needs (the multi-pick rounds would need it twice: partials, then verdicts).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcu_handoff.hip -o tools/micro/xcu_handoff && tools/micro/xcu_handoff
// Each group's slot and counter sit in 256-byte blocks of their own (no false sharing between groups).
// Every spin is bounded: a partner that does not show up within LIMIT polls raises `fail` and everybody leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024)
handoff(unsigned long long* slot, unsigned* cnt, int G, int stride, int rounds, int work, int* fail, unsigned long long* sink)
{
    // partners of a group: blocks `stride` apart (stride 8: same blockIdx % 8, observed to share an XCD; stride 1: neighbours,
    // observed to sit on different XCDs)
    const int member = (blockIdx.x / stride) % G;
    const int group = (blockIdx.x / (stride * G)) * stride + blockIdx.x % stride;
    __shared__ unsigned long long win;
    __shared__ int stop;
    unsigned long long acc = 0;
    float f = (float)threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        for (int w = 0; w < work; ++w) f = fmaf(f, 1.0000001f, 0.5f);          // stand-in for the round's own work
        if (threadIdx.x == 0) {
            const unsigned long long key = ((unsigned long long)(r + 1) << 40) | ((unsigned long long)((member * 7 + r) & 0xffff) << 8) | (unsigned)member;
            __hip_atomic_fetch_max(&slot[group * 32], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&cnt[group * 64], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(r + 1);
            int polls = 0;
            while (__hip_atomic_load(&cnt[group * 64], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > (1 << 22) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            win = __hip_atomic_load(&slot[group * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stop = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        acc += win;
        const int st = stop;                                                     // the same value in every thread
        __syncthreads();
        if (st) break;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc + (unsigned long long)f;
}

int main()
{
    const int rounds = 2000;
    unsigned long long *slot, *sink; unsigned* cnt; int* fail;
    CHECK(hipMalloc(&slot, 4096 * 8 * 32)); CHECK(hipMalloc(&sink, 4096 * 8)); CHECK(hipMalloc(&cnt, 4096 * 4 * 64)); CHECK(hipMalloc(&fail, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("threads G stride groups work | us per round (min of 5 launches of %d rounds)\n", rounds);
    const int threads_l[] = {64, 1024};
    for (int threads : threads_l)
        for (int G : {1, 2, 4, 8})
            for (int stride : {1, 8})
                for (int groups : {1, 8, 32})
                    for (int work : {0}) {
                        if (G == 1 && stride == 8) continue;
                        const int nblk = ((groups + stride - 1) / stride) * stride * G;   // whole strides of groups
                        if (nblk > 256) continue;                                             // one workgroup per CU: all co-resident
                        float best = 1e30f; int failed = 0;
                        for (int rep = 0; rep < 5; ++rep) {
                            CHECK(hipMemset(slot, 0, 4096 * 8 * 32)); CHECK(hipMemset(cnt, 0, 4096 * 4 * 64)); CHECK(hipMemset(fail, 0, 4));
                            CHECK(hipEventRecord(e0));
                            // 96 KB of dynamic LDS: a second workgroup never shares the CU
                            hipLaunchKernelGGL(handoff, dim3(nblk), dim3(threads), 96 * 1024, 0, slot, cnt, G, stride, rounds, work, fail, sink);
                            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                            int f; CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); failed |= f;
                            if (ms < best) best = ms;
                        }
                        printf("%7d %d %6d %6d %4d | %.3f%s\n", threads, G, stride, nblk / G, work, best * 1000.f / rounds, failed ? "  (a spin timed out)" : "");
                        fflush(stdout);
                        if (failed) return 2;
                    }
    return 0;
}
