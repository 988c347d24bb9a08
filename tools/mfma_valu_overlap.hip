// Developer probe: do vector-ALU instructions overlap with matrix instructions on a gfx950 SIMD?
//   leg "same":  one wave per SIMD; K independent v_max_f32 after every v_mfma_f32_32x32x16_bf16
//   leg "split": two waves per SIMD; waves 0-3 of the workgroup issue only matrix instructions, waves 4-7 only vector
//                instructions (K per matrix instruction of the partner), each wave timed by itself; PRIO sets
//                s_setprio 3 in the vector waves (1), in the matrix waves (2) or nowhere (0); SWAP makes the vector
//                waves the older ones (waves 0-3)
// Prints shader cycles per matrix instruction for the matrix wave and per K vector instructions for the vector wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o var_libs/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K>
__device__ __forceinline__ void valu_ops(float (&v)[8], float x)
{
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(v[j & 7]), "v"(x));
}

template <int K>
__global__ void same_kernel(const float *in, float *out, int iters, unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(lane + j) & 255]; b[j] = (__bf16)in[(lane * 3 + j) & 255]; }
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; ++j) { acc0[j] = in[j]; acc1[j] = in[j + 16]; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[j + 32];
    const float x = in[lane & 255];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            valu_ops<K>(v, x);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            valu_ops<K>(v, x);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int K, int PRIO, bool SWAP>
__global__ __launch_bounds__(512) void split_kernel(const float *in, float *out, int iters, unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool matrix_wave = SWAP ? wave >= 4 : wave < 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(lane + j) & 255]; b[j] = (__bf16)in[(lane * 3 + j) & 255]; }
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; ++j) { acc0[j] = in[j]; acc1[j] = in[j + 16]; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[j + 32];
    const float x = in[lane & 255];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (matrix_wave) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        }
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) valu_ops<K>(v, x);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) cycles[wave == 0 ? 0 : 1] = t1 - t0;
}

static const int ITERS = 20000;
template <int K>
static void run_same(const float *in, float *out, unsigned long long *d, int cus)
{
    same_kernel<K><<<cus, 256>>>(in, out, ITERS, d);
    CK(hipDeviceSynchronize());
    unsigned long long c[2]; CK(hipMemcpy(c, d, 16, hipMemcpyDeviceToHost));
    std::printf("same  K=%d: %.1f cycles per matrix instruction (32 = the pipe alone; %d = no overlap at 4 cycles per vector instruction)\n", K,
                (double)c[0] / (ITERS * 8.0), 32 + 4 * K);
}
template <int K, int PRIO, bool SWAP>
static void run_split(const float *in, float *out, unsigned long long *d, int cus)
{
    split_kernel<K, PRIO, SWAP><<<cus, 512>>>(in, out, ITERS, d);
    CK(hipDeviceSynchronize());
    unsigned long long c[2]; CK(hipMemcpy(c, d, 16, hipMemcpyDeviceToHost));
    const double m = (double)(SWAP ? c[1] : c[0]) / (ITERS * 8.0), v = (double)(SWAP ? c[0] : c[1]) / (ITERS * 8.0);
    std::printf("split K=%d prio=%d %s: matrix wave %.1f cycles per matrix instruction, vector wave %.1f cycles per %d vector instructions (alone: 32 and %d)\n",
                K, PRIO, SWAP ? "vector waves older" : "matrix waves older", m, v, K, 4 * K);
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float *in, *out; unsigned long long *d;
    CK(hipMalloc(&in, 1024)); CK(hipMemset(in, 0, 1024));
    CK(hipMalloc(&out, sizeof(float) * cus * 512)); CK(hipMalloc(&d, 16));
    run_same<0>(in, out, d, cus); run_same<0>(in, out, d, cus); run_same<2>(in, out, d, cus); run_same<4>(in, out, d, cus);
    run_same<6>(in, out, d, cus); run_same<8>(in, out, d, cus); run_same<12>(in, out, d, cus);
    run_split<4, 0, false>(in, out, d, cus); run_split<4, 1, false>(in, out, d, cus); run_split<4, 2, false>(in, out, d, cus);
    run_split<4, 0, true>(in, out, d, cus); run_split<4, 1, true>(in, out, d, cus); run_split<4, 2, true>(in, out, d, cus);
    run_split<8, 0, false>(in, out, d, cus); run_split<8, 1, false>(in, out, d, cus); run_split<8, 0, true>(in, out, d, cus);
    run_split<6, 0, false>(in, out, d, cus); run_split<6, 0, true>(in, out, d, cus);
    return 0;
}
