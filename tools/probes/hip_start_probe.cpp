// how long does the HIP runtime take to start in a process that holds no kernels at all?  (compare with the CLI's
// "problem_create (HIP runtime start)" phase: the difference is what the library's own code object costs)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    hipGetDeviceCount(&n);
    hipSetDevice(0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *p = nullptr;
    hipMalloc(&p, 1 << 20);
    auto t1 = std::chrono::steady_clock::now();
    std::printf("bare HIP start (device count, set device, stream, 1 MiB malloc): %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
