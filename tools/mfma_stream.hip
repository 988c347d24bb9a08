// Developer probe: what in the second scoring kernel's per-wave stream keeps one wave from feeding the matrix pipe alone?
// One wave per SIMD runs "blocks" of two phases of eight dependent v_mfma_f32_32x32x16_bf16 each (phase A into acc0, phase B
// into acc1), with the kernel's other instructions added one kind at a time:
//   bit 0: two v_max per matrix instruction on the other accumulator (the group maxima)
//   bit 1: one ds_read_b128 per matrix instruction of phase B into the fragment just used, waited for where phase A uses it
//   bit 2: v_cmp + scalar branch (never taken) at the end of each phase
//   bit 3: v_readfirstlane + s_mov m0 + global_load_lds_dwordx4 once per block
//   bit 4: the first matrix instruction of a phase takes its accumulator from a third register set (the biases)
// Prints shader cycles per matrix instruction (32 = the pipe).   Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_stream.hip -o var_libs/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int F>
__global__ __launch_bounds__(256) void stream_kernel(const float *in, float *out, const char *gsrc, int iters, unsigned long long *cycles)
{
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float *>(lds)[i] = in[i & 255];
    bf16x8 af[8], pb0[8], pb1[8];
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 8; ++j) { af[k][j] = (__bf16)in[(lane + j + k) & 255]; pb0[k][j] = (__bf16)in[(lane * 3 + j + k) & 255]; pb1[k][j] = (__bf16)in[(lane * 5 + j + k) & 255]; }
    f32x16 acc0, acc1, bv;
    for (int j = 0; j < 16; ++j) { acc0[j] = in[j]; acc1[j] = in[j + 16]; bv[j] = in[j + 32]; }
    float g0 = 0.f, g1 = 0.f, th = in[7] + 1e30f;
    const int uj = lane & 31, hh = lane >> 5;
    int a_off[8];
    for (int k = 0; k < 8; ++k) a_off[k] = uj * 256 + ((((uj & 15) ^ hh) * 16) ^ (32 * k));   // the kernel's swizzled fragment addresses: conflict-free
    const char *frag = lds;
    const uint32_t ldsdst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + 32768 + wave * 1024;
    const char *src = gsrc + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k], pb0[k], (F & 16) && k == 0 ? bv : acc0, 0, 0, 0);
            if (F & 1) { if (k >= 1 && k <= 4) g1 = fmaxf(g1, fmaxf(fmaxf(acc1[4 * (k - 1)], acc1[4 * (k - 1) + 1]), fmaxf(acc1[4 * (k - 1) + 2], acc1[4 * (k - 1) + 3]))); }
            if ((F & 8) && k == 6) {
                const uint32_t d = __builtin_amdgcn_readfirstlane(ldsdst + (uint32_t)((i & 7) * 4096));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory", "m0");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (F & 4) { if (__ballot(g1 > th) != 0) [[unlikely]] { out[threadIdx.x] = g1; g1 = 0.f; } }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k], pb1[k], (F & 16) && k == 0 ? bv : acc1, 0, 0, 0);
            if (F & 2) af[k] = *reinterpret_cast<const bf16x8 *>(frag + ((i & 3) * 8192) + a_off[k]);
            if (F & 1) { if (k >= 2 && k <= 5) g0 = fmaxf(g0, fmaxf(fmaxf(acc0[4 * (k - 2)], acc0[4 * (k - 2) + 1]), fmaxf(acc0[4 * (k - 2) + 2], acc0[4 * (k - 2) + 3]))); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (F & 4) { if (__ballot(g0 > th) != 0) [[unlikely]] { out[threadIdx.x + 256] = g0; g0 = 0.f; } }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = g0 + g1;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    for (int k = 0; k < 8; ++k) s += (float)af[k][0];
    out[blockIdx.x * 256 + threadIdx.x + 512] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int F>
static void run(const float *in, float *out, const char *gsrc, unsigned long long *d, int cus)
{
    const int iters = 20000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    stream_kernel<F><<<cus, 256, 131072>>>(in, out, gsrc, iters, d);
    CK(hipDeviceSynchronize());
    unsigned long long c; CK(hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost));
    std::printf("features %2d (%s%s%s%s%s): %.1f cycles per matrix instruction\n", F, F & 1 ? "max " : "", F & 2 ? "ds_read " : "", F & 4 ? "branch " : "",
                F & 8 ? "glds " : "", F & 16 ? "bias-C " : "", (double)c / (iters * 16.0));
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float *in, *out; unsigned long long *d; char *gsrc;
    CK(hipMalloc(&in, 1024)); CK(hipMemset(in, 0, 1024));
    CK(hipMalloc(&out, sizeof(float) * (cus * 256 + 1024))); CK(hipMalloc(&d, 8));
    CK(hipMalloc(&gsrc, (size_t)cus * 256 * 16)); CK(hipMemset(gsrc, 0, (size_t)cus * 256 * 16));
    run<0>(in, out, gsrc, d, cus); run<1>(in, out, gsrc, d, cus); run<2>(in, out, gsrc, d, cus); run<3>(in, out, gsrc, d, cus);
    run<4>(in, out, gsrc, d, cus); run<7>(in, out, gsrc, d, cus); run<8>(in, out, gsrc, d, cus); run<15>(in, out, gsrc, d, cus);
    run<16>(in, out, gsrc, d, cus); run<31>(in, out, gsrc, d, cus);
    return 0;
}
