// Developer probe: times rocPRIM's onesweep radix sort of (key, value) pairs under different
// kernel configurations at the size and key width of one training batch.
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class Cfg>
static void run(const char *name, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits)
{
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, (hipStream_t)0));
    void *tmp = nullptr;
    CK(hipMalloc(&tmp, bytes + 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, (hipStream_t)0));
    CK(hipEventRecord(a, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, (hipStream_t)0));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s n=%zu bits=%d: %.1f us\n", name, n, bits, ms * 1e3f / reps);
    CK(hipFree(tmp));
}

template <unsigned BS, unsigned IPT, unsigned RB>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, RB>, 4096>;

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3u << 20;
    const int bits = argc > 2 ? atoi(argv[2]) : 21;
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)(s >> 20) & ((1u << bits) - 1u); hv[i] = (uint32_t)i; }
    uint32_t *kin, *kout, *vin, *vout;
    CK(hipMalloc(&kin, 4 * n)); CK(hipMalloc(&kout, 4 * n)); CK(hipMalloc(&vin, 4 * n)); CK(hipMalloc(&vout, 4 * n));
    CK(hipMemcpy(kin, hk.data(), 4 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(vin, hv.data(), 4 * n, hipMemcpyHostToDevice));
    run<rocprim::default_config>("default", kin, kout, vin, vout, n, bits);
    run<OS<256, 8, 8>>("256x8 r8", kin, kout, vin, vout, n, bits);
    run<OS<256, 4, 8>>("256x4 r8", kin, kout, vin, vout, n, bits);
    run<OS<256, 16, 8>>("256x16 r8", kin, kout, vin, vout, n, bits);
    run<OS<256, 8, 7>>("256x8 r7", kin, kout, vin, vout, n, bits);
    run<OS<256, 12, 7>>("256x12 r7", kin, kout, vin, vout, n, bits);
    run<OS<512, 8, 7>>("512x8 r7", kin, kout, vin, vout, n, bits);
    run<OS<256, 6, 8>>("256x6 r8", kin, kout, vin, vout, n, bits);
    run<OS<256, 16, 7>>("256x16 r7", kin, kout, vin, vout, n, bits);
    run<OS<256, 8, 6>>("256x8 r6", kin, kout, vin, vout, n, bits);
    return 0;
}
