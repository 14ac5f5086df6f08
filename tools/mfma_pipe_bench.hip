// What does the bf16 matrix pipe deliver when nothing else is in the way?  (measurement aid, not part of the library)
// Register-only loops of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 on independent accumulators, one or two
// wavefronts per SIMD on every CU, long enough (several ms) for the clock to settle.  Printed: TFLOP/s, the fraction of
// the nominal dense peak (2.5 PFLOP/s = 1,024 flop per SIMD and cycle at 2.4 GHz), and the shader clock the run held
// (s_memtime ticks per wall-clock microsecond).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/mfma_pipe_bench.bin tools/mfma_pipe_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RND: four different operand pairs of pseudo-random bit patterns (exponents near 1) in turn instead of one pair of small integers
template <int NACC, bool RND = false>
__global__ __launch_bounds__(512) void k_pipe16(int iters, float* out, unsigned long long* ticks) {
    f32x4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + 12345u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (RND) {
                h = h * 1664525u + 1013904223u; a[r][k] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00u | ((h >> 9) & 0x83ffu)));
                h = h * 1664525u + 1013904223u; b[r][k] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00u | ((h >> 9) & 0x83ffu)));
            } else { a[r][k] = (__bf16)(float)(threadIdx.x & 3); b[r][k] = (__bf16)(float)((threadIdx.x >> 2) & 3); }
        }
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RND ? r : 0], b[RND ? (r + k) & 3 : 0], acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < NACC; ++k) s += acc[k];
    if (s[0] == 123.456f) out[threadIdx.x] = s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512) void k_pipe32(int iters, float* out, unsigned long long* ticks) {
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(threadIdx.x & 3); b[k] = (__bf16)(float)((threadIdx.x >> 2) & 3); }
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][15];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int threads, int nacc, double flop_per_mfma, float* out, unsigned long long* ticks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, threads>>>(iters / 10, out, ticks);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<256, threads>>>(iters, out, ticks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t = 0;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double mfmas = (double)iters * 4 * nacc * (threads / 64) * 256;
    const double tf = mfmas * flop_per_mfma / (ms * 1e-3) * 1e-12;
    printf("%-34s %d wavefront(s) per SIMD, %d accumulators: %7.3f ms  %7.1f TFLOP/s = %.2f of 2,500; %6.1f s_memtime ticks per us; %5.1f ticks per MFMA and SIMD\n", name, threads / 256, nacc,
           ms, tf, tf / 2500.0, (double)t / (ms * 1e3), (double)t / ((double)iters * 4 * nacc * (threads / 256)));
}

int main() {
    float* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&ticks, 64);
    run("v_mfma_f32_16x16x32_bf16", k_pipe16<4>, 256, 4, 16384.0, out, ticks);
    run("v_mfma_f32_16x16x32_bf16", k_pipe16<8>, 256, 8, 16384.0, out, ticks);
    run("v_mfma_f32_16x16x32_bf16", k_pipe16<8>, 512, 8, 16384.0, out, ticks);
    run("v_mfma_f32_16x16x32_bf16 random", k_pipe16<8, true>, 256, 8, 16384.0, out, ticks);
    run("v_mfma_f32_16x16x32_bf16 random", k_pipe16<8, true>, 512, 8, 16384.0, out, ticks);
    run("v_mfma_f32_32x32x16_bf16", k_pipe32<4>, 256, 4, 32768.0, out, ticks);
    run("v_mfma_f32_32x32x16_bf16", k_pipe32<4>, 512, 4, 32768.0, out, ticks);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
    return 0;
}
