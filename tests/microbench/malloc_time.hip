// How long hipMalloc / hipFree take as a function of size, and the virtual-memory API beside them (diagnostics, GPU box):
//   hipcc --offload-arch=gfx950 -O2 tests/microbench/malloc_time.hip -o /tmp/malloc_time && /tmp/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main()
{
    hipSetDevice(0);
    hipFree(0);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    for (int rep = 0; rep < 2; rep++)
        for (size_t gb : {1, 8, 32, 64, 128, 240}) {
            const size_t n = gb << 30;
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, n);
            double t1 = now();
            if (e != hipSuccess) { printf("hipMalloc %zu GB failed: %s\n", gb, hipGetErrorString(e)); continue; }
            touch<<<(unsigned)((n / (2 << 20) + 255) / 256), 256>>>((char *)p, n, 2 << 20);
            hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("rep %d: hipMalloc %4zu GB: %.3f s, first touch (1 B per 2 MiB) %.3f s, hipFree %.3f s\n", rep, gb, t1 - t0, t2 - t1, t3 - t2);
        }
    // virtual memory API: reserve the range at once, back it granule by granule
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("vmm granularity %zu (%s)\n", gran, hipGetErrorString(e));
        const size_t total = 128ull << 30, piece = 2ull << 30;
        void *va = nullptr;
        double t0 = now();
        e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
        double t1 = now();
        printf("reserve 128 GB: %.4f s (%s)\n", t1 - t0, hipGetErrorString(e));
        if (e == hipSuccess) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            double tc = 0, tm = 0;
            for (size_t off = 0; off < total; off += piece) {
                hipMemGenericAllocationHandle_t h;
                double a = now();
                e = hipMemCreate(&h, piece, &prop, 0);
                double b = now();
                if (e != hipSuccess) { printf("hipMemCreate failed at %zu GB: %s\n", off >> 30, hipGetErrorString(e)); break; }
                e = hipMemMap((char *)va + off, piece, 0, h, 0);
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                if (e == hipSuccess) e = hipMemSetAccess((char *)va + off, piece, &acc, 1);
                double c = now();
                if (e != hipSuccess) { printf("map failed: %s\n", hipGetErrorString(e)); break; }
                tc += b - a; tm += c - b;
                hs.push_back(h);
            }
            printf("vmm: %zu pieces of 2 GB: create %.3f s, map+access %.3f s\n", hs.size(), tc, tm);
            double a = now();
            touch<<<(unsigned)((hs.size() * piece / (2 << 20) + 255) / 256), 256>>>((char *)va, hs.size() * piece, 2 << 20);
            e = hipDeviceSynchronize();
            printf("vmm touch: %.3f s (%s)\n", now() - a, hipGetErrorString(e));
            for (size_t i = 0; i < hs.size(); i++) { hipMemUnmap((char *)va + i * piece, piece); hipMemRelease(hs[i]); }
            hipMemAddressFree(va, total);
        }
    }
    return 0;
}
