// Host <-> device copy rates on the box (pageable vs pinned, CPU staging copy): what a
// file -> HBM -> file pipeline can hope for.  hipcc -O2 -o pcie_probe pcie_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char *d, const char *s, size_t n, int threads) {
    std::vector<std::thread> t;
    size_t per = (n + threads - 1) / threads;
    for (int i = 0; i < threads; ++i) {
        size_t lo = i * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) t.emplace_back([=] { memcpy(d + lo, s + lo, hi - lo); });
    }
    for (auto &x : t) x.join();
}
int main() {
    const size_t n = 1ull << 30;
    char *pageable = (char *)malloc(n), *pageable2 = (char *)malloc(n), *pinned, *dev;
    memset(pageable, 1, n); memset(pageable2, 2, n);
    hipHostMalloc((void **)&pinned, n); memset(pinned, 3, n);
    hipMalloc((void **)&dev, n);
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); double a = now() - t;
        t = now(); hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice); double b = now() - t;
        t = now(); hipMemcpy(pageable2, dev, n, hipMemcpyDeviceToHost); double c = now() - t;
        t = now(); hipMemcpy(pinned, dev, n, hipMemcpyDeviceToHost); double d = now() - t;
        printf("rep %d: H2D pageable %.1f GB/s, pinned %.1f GB/s; D2H pageable %.1f GB/s, pinned %.1f GB/s\n", rep, n / a / 1e9, n / b / 1e9, n / c / 1e9, n / d / 1e9);
    }
    for (int th : {1, 2, 4, 8, 16}) {
        double t = now(); par_copy(pinned, pageable, n, th); double a = now() - t;
        t = now(); par_copy(pageable2, pinned, n, th); double b = now() - t;
        printf("CPU copy %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s\n", th, n / a / 1e9, n / b / 1e9);
    }
    double t = now(); hipHostRegister(pageable, n, hipHostRegisterDefault); double a = now() - t;
    t = now(); hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); double b = now() - t;
    t = now(); hipHostUnregister(pageable); double c = now() - t;
    printf("hipHostRegister 1 GiB: %.1f ms, then H2D %.1f GB/s, unregister %.1f ms\n", a * 1e3, n / b / 1e9, c * 1e3);
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
