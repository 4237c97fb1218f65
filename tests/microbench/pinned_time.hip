// What host memory for a device-to-host copy costs (diagnostics, GPU box): hipHostMalloc, hipHostRegister of malloc'd and of
// huge-page-advised memory, and the copies themselves into pinned and pageable memory; and what starting HIP costs.
//   hipcc --offload-arch=gfx950 -O2 tests/microbench/pinned_time.hip -o /tmp/pinned_time && /tmp/pinned_time [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 400;
    const size_t n = mb << 20;
    double t0 = now();
    hipSetDevice(0);
    hipFree(0);
    double t1 = now();
    printf("hipSetDevice + hipFree(0): %.1f ms\n", (t1 - t0) * 1e3);
    hipStream_t st;
    t0 = now();
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    t1 = now();
    printf("first hipStreamCreate: %.1f ms\n", (t1 - t0) * 1e3);
    char *d = nullptr;
    t0 = now();
    hipMalloc((void **)&d, n);
    hipMemsetAsync(d, 1, n, st);
    hipStreamSynchronize(st);
    t1 = now();
    printf("hipMalloc + memset %zu MB: %.1f ms\n", mb, (t1 - t0) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        char *h = nullptr;
        t0 = now();
        hipHostMalloc((void **)&h, n, hipHostMallocDefault);
        t1 = now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        double t2 = now();
        hipHostFree(h);
        double t3 = now();
        printf("rep %d: hipHostMalloc %.1f ms, D2H %.1f ms (%.1f GB/s), hipHostFree %.1f ms\n", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
        t0 = now();
        hipHostMalloc((void **)&h, n, hipHostMallocNonCoherent | hipHostMallocPortable);
        t1 = now();
        hipHostFree(h);
        printf("rep %d: hipHostMalloc(non-coherent) %.1f ms\n", rep, (t1 - t0) * 1e3);
        // pageable
        t0 = now();
        h = (char *)malloc(n);
        memset(h, 0, n);
        t1 = now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        t2 = now();
        printf("rep %d: malloc + first touch %.1f ms, D2H into pageable (touched) %.1f ms (%.1f GB/s)\n", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
        free(h);
        t0 = now();
        h = (char *)malloc(n);
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        t1 = now();
        printf("rep %d: malloc + D2H into untouched pageable %.1f ms (%.1f GB/s)\n", rep, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        free(h);
        // register malloc'd memory
        t0 = now();
        h = (char *)aligned_alloc(2 << 20, n);
        t1 = now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
        t2 = now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        t3 = now();
        printf("rep %d: hipHostRegister(untouched malloc) %.1f ms (%s), D2H %.1f ms\n", rep, (t2 - t1) * 1e3, hipGetErrorString(e), (t3 - t2) * 1e3);
        hipHostUnregister(h);
        free(h);
        // huge pages
        t0 = now();
        h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(h, n, MADV_HUGEPAGE);
        t1 = now();
        e = hipHostRegister(h, n, hipHostRegisterDefault);
        t2 = now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        t3 = now();
        printf("rep %d: hipHostRegister(mmap + MADV_HUGEPAGE) %.1f ms (%s), D2H %.1f ms\n", rep, (t2 - t1) * 1e3, hipGetErrorString(e), (t3 - t2) * 1e3);
        hipHostUnregister(h);
        munmap(h, n);
        // several threads' worth: chunks of 32 MB allocated one by one (what a background allocator would do)
        t0 = now();
        char *hs[64];
        size_t k = 0;
        for (size_t o = 0; o < n && k < 64; o += 32 << 20) hipHostMalloc((void **)&hs[k++], 32 << 20, hipHostMallocDefault);
        t1 = now();
        for (size_t i = 0; i < k; i++) hipHostFree(hs[i]);
        printf("rep %d: %zu x hipHostMalloc(32 MB) %.1f ms\n", rep, k, (t1 - t0) * 1e3);
    }
    return 0;
}
