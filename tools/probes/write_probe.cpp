#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
using clk = std::chrono::steady_clock;
int main(int argc, char **argv) {
    const int mode = atoi(argv[1]), T = atoi(argv[2]);
    const size_t bytes = (size_t)564 << 20, chunk = (size_t)16 << 20;
    std::vector<char> src(chunk, 'x');
    auto t0 = clk::now();
    int fd = open(argv[3], O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (ftruncate(fd, bytes)) return 1;
    char *map = nullptr;
    if (mode == 1) map = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    std::atomic<size_t> next{0};
    const size_t n = (bytes + chunk - 1) / chunk;
    auto work = [&]() {
        for (size_t k = next++; k < n; k = next++) {
            size_t off = k * chunk, len = std::min(chunk, bytes - off);
            if (mode == 0) { if (pwrite(fd, src.data(), len, off) != (ssize_t)len) abort(); }
            else memcpy(map + off, src.data(), len);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (map) munmap(map, bytes);
    close(fd);
    printf("mode %d threads %d: %.1f ms\n", mode, T, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
}
