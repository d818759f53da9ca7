// d2hbench.hip -- device-to-host copy rate by copy size, host allocation flavour and number of streams.
// Diagnostic for the host-pointer path's pipeline (chunk size, one or two copy streams).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MAXB = 1ull << 30;
    uint8_t *d;
    CK(hipMalloc(&d, MAXB));
    CK(hipMemset(d, 1, MAXB));
    struct Flavour { const char *name; unsigned flags; } fl[] = {
        {"default", hipHostMallocDefault}, {"noncoherent", hipHostMallocNonCoherent}, {"coherent", hipHostMallocCoherent},
        {"numa_user", hipHostMallocNumaUser}};
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (auto &f : fl) {
        uint8_t *h;
        if (hipHostMalloc(reinterpret_cast<void **>(&h), MAXB, f.flags) != hipSuccess) { printf("%s: alloc failed\n", f.name); (void)hipGetLastError(); continue; }
        for (size_t i = 0; i < MAXB; i += 4096) h[i] = 0;  // touch
        for (size_t mb : {1, 4, 20, 44, 99, 256, 1024}) {
            const size_t bytes = mb << 20;
            const int reps = (int)(MAXB / bytes) < 4 ? 4 : (int)(MAXB / bytes);
            for (int ns = 1; ns <= 2; ++ns) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int i = 0; i < reps; ++i) {
                    const size_t off = ((size_t)i * bytes) % (MAXB - bytes + 1);
                    if (ns == 1) CK(hipMemcpyAsync(h + off, d + off, bytes, hipMemcpyDeviceToHost, s[0]));
                    else {  // the same bytes split over two streams
                        CK(hipMemcpyAsync(h + off, d + off, bytes / 2, hipMemcpyDeviceToHost, s[0]));
                        CK(hipMemcpyAsync(h + off + bytes / 2, d + off + bytes / 2, bytes - bytes / 2, hipMemcpyDeviceToHost, s[1]));
                    }
                }
                CK(hipDeviceSynchronize());
                const double dt = now() - t0;
                printf("%-12s %5zu MB  streams %d  %7.2f GB/s\n", f.name, mb, ns, reps * (double)bytes / dt / 1e9);
            }
        }
        CK(hipHostFree(h));
    }
    // D2H while a kernel-free H2D runs the other way (the pipeline uploads inputs once, so this is only for scale)
    return 0;
}
