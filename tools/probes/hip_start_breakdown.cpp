// where the HIP runtime's start goes, call by call, in a process that holds no kernels (the CLI's "HIP runtime start" phase
// is 120-260 ms of every command: 30-60 % of `synthetic --blocks 128` since r04).  Build + run on the GPU box:
//   hipcc -O2 tools/probes/hip_start_breakdown.cpp -o /tmp/hsb && for i in 1 2 3; do /tmp/hsb; done
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    auto t = std::chrono::steady_clock::now();
    const auto t0 = t;
    hipInit(0);                                   std::printf("hipInit              %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    int n = 0; hipGetDeviceCount(&n);             std::printf("hipGetDeviceCount    %7.1f ms  (%d)\n", ms(t), n); t = std::chrono::steady_clock::now();
    hipSetDevice(0);                              std::printf("hipSetDevice         %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    void *p = nullptr; hipMalloc(&p, 1 << 20);    std::printf("first hipMalloc      %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                                                  std::printf("hipStreamCreate      %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    hipMemsetAsync(p, 0, 1 << 20, s); hipStreamSynchronize(s);
                                                  std::printf("first memset + sync  %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    void *h = nullptr; hipHostMalloc(&h, 32 << 20, hipHostMallocDefault);
                                                  std::printf("hipHostMalloc 32 MiB %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    void *q = nullptr; hipMalloc(&q, (size_t)1 << 30);
                                                  std::printf("hipMalloc 1 GiB      %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    hipMemcpy(q, h, 32 << 20, hipMemcpyHostToDevice);
                                                  std::printf("first H2D 32 MiB     %7.1f ms\n", ms(t)); t = std::chrono::steady_clock::now();
    std::printf("total                %7.1f ms\n", ms(t0));
    return 0;
}
