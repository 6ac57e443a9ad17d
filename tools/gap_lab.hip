// gap_lab.hip -- what a chain of tiny DEPENDENT kernels costs on this device: the floor of a "one launch per greedy round"
// form of the large path's chain.  Each kernel reads what the previous one wrote (64 KB through
// L2) and writes the next buffer: G workgroups of 512 threads.  Plain stream launches and the same chain as a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/gap_lab.hip -o /tmp/gap_lab && /tmp/gap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void step_kernel(const unsigned long long* in, unsigned long long* out, int n, int work) {
    extern __shared__ unsigned long long lds[];
    // every workgroup reads the whole previous array (as a round would: 8 192 bins), does `work` dependent LDS round trips,
    // and writes its own slice of the next one
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    const int per = n / gridDim.x;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        int g = blockIdx.x * per + i;
        unsigned long long v = lds[g];
        int p = g;
        for (int k = 0; k < work; ++k) p = (int)((lds[p] + (unsigned)k) % (unsigned)n);      // a dependent chain (a binary search's shape)
        out[g] = v + 1 + (unsigned long long)(p & 1 ? 0 : 0);
    }
}

// The other form: ONE persistent kernel of G workgroups, a software grid barrier (agent-scope counter) between rounds.
__global__ __launch_bounds__(512) void persistent_kernel(unsigned long long* a, unsigned long long* b, int n, int rounds, int work,
                                                        unsigned int* counter) {
    extern __shared__ unsigned long long lds[];
    const int per = n / gridDim.x;
    for (int q = 0; q < rounds; ++q) {
        const unsigned long long* in = (q & 1) ? b : a;
        unsigned long long* out = (q & 1) ? a : b;
        for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = __builtin_nontemporal_load(in + i);
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            int g = blockIdx.x * per + i;
            unsigned long long v = lds[g];
            int p = g;
            for (int k = 0; k < work; ++k) p = (int)((lds[p] + (unsigned)k) % (unsigned)n);
            out[g] = v + 1 + (unsigned long long)(p & 1 ? 0 : 0);
        }
        // grid barrier: everybody's stores visible at agent scope, then count in and wait for the round's total
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = (unsigned int)(q + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main() {
    const int n = 8192, rounds = 128;
    unsigned long long *a, *b;
    CHECK(hipMalloc(&a, n * 8)); CHECK(hipMalloc(&b, n * 8));
    CHECK(hipMemset(a, 0, n * 8)); CHECK(hipMemset(b, 0, n * 8));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int G : {1, 4, 16, 64}) {
        for (int work : {0, 13, 40}) {
            auto chain = [&](hipStream_t s) {
                for (int q = 0; q < rounds; ++q)
                    hipLaunchKernelGGL(step_kernel, dim3(G), dim3(512), n * 8, s, (q & 1) ? b : a, (q & 1) ? a : b, n, work);
            };
            chain(st); CHECK(hipStreamSynchronize(st));
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0, st)); chain(st); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            // the same chain as a graph
            hipGraph_t graph; hipGraphExec_t exec;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); chain(st); CHECK(hipStreamEndCapture(st, &graph));
            CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CHECK(hipGraphLaunch(exec, st)); CHECK(hipStreamSynchronize(st));
            float gbest = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0, st)); CHECK(hipGraphLaunch(exec, st)); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < gbest) gbest = ms;
            }
            CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
            printf("G = %2d workgroups, %2d dependent LDS trips per element: %d kernels in a stream %.3f ms = %.2f us each; as a hipGraph %.3f ms = %.2f us each\n",
                   G, work, rounds, best, best * 1e3f / rounds, gbest, gbest * 1e3f / rounds);
        }
    }
    unsigned int* counter; CHECK(hipMalloc(&counter, 4));
    for (int G : {2, 4, 8, 16, 32, 64}) {
        for (int work : {0, 13, 40}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipMemsetAsync(counter, 0, 4, st));
                CHECK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(512), n * 8, st, a, b, n, rounds, work, counter);
                CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
            }
            printf("persistent, G = %2d workgroups, %2d dependent LDS trips per element: %d rounds %.3f ms = %.2f us per round (grid barrier + 64 KB from L2 + the trips)\n",
                   G, work, rounds, best, best * 1e3f / rounds);
        }
    }
    return 0;
}
