// graph_overhead.hip -- what a hipGraph replay costs on this runtime as a function of its node count: k tiny kernels (+ an H2D
// and a D2H copy node, as the calibration graph has) captured once, replayed 2000 times, host time per hipGraphLaunch +
// hipStreamSynchronize; and the same k kernels queued directly on the stream.
// hipcc --offload-arch=gfx950 -O3 tools/r04/graph_overhead.hip -o tools/r04/graph_overhead
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void tiny(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *d, *hin, *hout;
    CHECK(hipMalloc(&d, 1 << 20));
    CHECK(hipHostMalloc(&hin, 4096, hipHostMallocDefault));
    CHECK(hipHostMalloc(&hout, 4096, hipHostMallocDefault));
    for (int copies = 0; copies <= 1; ++copies)
        for (int k : {1, 2, 3, 4, 6, 8}) {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            if (copies) CHECK(hipMemcpyAsync(d, hin, 512, hipMemcpyHostToDevice, st));
            for (int i = 0; i < k; ++i) tiny<<<64, 256, 0, st>>>(d, 64 * 256);
            if (copies) CHECK(hipMemcpyAsync(hout, d, 512, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamEndCapture(st, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            std::vector<double> tg, td;
            for (int rep = 0; rep < 2200; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                CHECK(hipGraphLaunch(ge, st));
                CHECK(hipStreamSynchronize(st));
                auto t1 = std::chrono::steady_clock::now();
                if (copies) CHECK(hipMemcpyAsync(d, hin, 512, hipMemcpyHostToDevice, st));
                for (int i = 0; i < k; ++i) tiny<<<64, 256, 0, st>>>(d, 64 * 256);
                if (copies) CHECK(hipMemcpyAsync(hout, d, 512, hipMemcpyDeviceToHost, st));
                CHECK(hipStreamSynchronize(st));
                auto t2 = std::chrono::steady_clock::now();
                if (rep >= 200) {
                    tg.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                    td.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
                }
            }
            std::sort(tg.begin(), tg.end()); std::sort(td.begin(), td.end());
            printf("{\"kernels\": %d, \"copy_nodes\": %d, \"graph_replay_us_median\": %.2f, \"direct_queue_us_median\": %.2f}\n", k, 2 * copies,
                   tg[tg.size() / 2], td[td.size() / 2]);
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
        }
    return 0;
}
