// Developer probe: does s_setprio decide which of two waves of a SIMD gets the matrix pipe?  Workgroups of eight waves
// (two per SIMD), every wave issues the same chain of matrix instructions; waves 4-7 (the younger ones) optionally raise
// their priority.  Prints the cycles wave 0 and wave 4 of one workgroup took (alone on the SIMD: 32 per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_prio.hip -o var_libs/mfma_prio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PRIO_YOUNG, int PRIO_OLD>
__global__ __launch_bounds__(512) void prio_kernel(const float *in, float *out, int iters, unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(lane + j) & 255]; b[j] = (__bf16)in[(lane * 3 + j) & 255]; }
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; ++j) { acc0[j] = in[j]; acc1[j] = in[j + 16]; }
    __syncthreads();
    if (wave >= 4) __builtin_amdgcn_s_setprio(PRIO_YOUNG); else __builtin_amdgcn_s_setprio(PRIO_OLD);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) cycles[wave == 0 ? 0 : 1] = t1 - t0;
}

template <int PY, int PO>
static void run(const float *in, float *out, unsigned long long *d, int cus)
{
    const int iters = 4000;
    prio_kernel<PY, PO><<<cus, 512>>>(in, out, iters, d);
    CK(hipDeviceSynchronize());
    unsigned long long c[2]; CK(hipMemcpy(c, d, 16, hipMemcpyDeviceToHost));
    std::printf("priority old waves %d, young waves %d: wave 0 %.1f, wave 4 %.1f cycles per matrix instruction (16 per iteration)\n", PO, PY,
                c[0] / (iters * 16.0), c[1] / (iters * 16.0));
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float *in, *out; unsigned long long *d;
    CK(hipMalloc(&in, 1024)); CK(hipMemset(in, 0, 1024));
    CK(hipMalloc(&out, sizeof(float) * cus * 512)); CK(hipMalloc(&d, 16));
    run<0, 0>(in, out, d, cus); run<3, 0>(in, out, d, cus); run<0, 3>(in, out, d, cus); run<1, 0>(in, out, d, cus);
    return 0;
}
