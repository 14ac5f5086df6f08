// Weight-gradient product: the fp32 kernel (k_gemm_tn_w) beside the bf16-pieces kernel (k_gemm_tn_ws) on one problem of
// config 4's shape (H = 300, 10^6 rows, 256 slices), and both against a float64 product on a small problem.
// (measurement aid, not part of the library)
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/tn_split_bench.bin tools/tn_split_bench.hip
#include "../graph-convnet-uelb_amd/csrc/kernels_tn_bf.hpp"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void launch(int which, int H, const TnProblems& pr, int slices) {
    if (which == 0) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TnWideLds)));
        k_gemm_tn_w<<<dim3(slices, 1, 1), 512, sizeof(TnWideLds)>>>(H, pr);
    } else {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_ws), hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS));
        k_gemm_tn_ws<<<dim3(slices, 1, 1), TS_TH, TS_LDS>>>(H, pr);
    }
    CK(hipGetLastError());
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 300;
    const size_t big = argc > 2 ? (size_t)atol(argv[2]) : 1000000, small = 5000 + 7;
    const size_t slab = (size_t)H * H + H;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx(small * H), hy(small * H);
    for (auto& v : hx) v = nd(rng) * 1e-3f;
    for (auto& v : hy) v = nd(rng) + 0.5f;
    float *X, *Y, *out;
    CK(hipMalloc(&X, big * H * 4)); CK(hipMalloc(&Y, big * H * 4)); CK(hipMalloc(&out, 256 * slab * 4));
    CK(hipMemcpy(X, hx.data(), small * H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Y, hy.data(), small * H * 4, hipMemcpyHostToDevice));
    // float64 product of the small problem
    std::vector<double> ref(slab, 0.0);
    for (size_t r = 0; r < small; ++r)
        for (int n = 0; n < H; ++n) {
            const double x = hx[r * H + n];
            ref[(size_t)H * H + n] += x;
            for (int k = 0; k < H; ++k) ref[(size_t)n * H + k] += x * (double)hy[r * H + k];
        }
    double rmax = 0, bmax = 0;
    for (size_t i = 0; i < (size_t)H * H; ++i) rmax = std::max(rmax, std::fabs(ref[i]));
    for (size_t i = (size_t)H * H; i < slab; ++i) bmax = std::max(bmax, std::fabs(ref[i]));
    for (int which = 0; which < 2; ++which) {
        const int slices = 37;
        TnProblems pr = {};
        pr.X[0] = X; pr.Y[0] = Y; pr.rows[0] = small; pr.nslices[0] = slices; pr.out[0] = out; pr.out_stride[0] = slab;
        CK(hipMemset(out, 0xff, 256 * slab * 4));
        launch(which, H, pr, slices);
        CK(hipDeviceSynchronize());
        std::vector<float> ho((size_t)slices * slab);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double err = 0, berr = 0;
        for (size_t i = 0; i < slab; ++i) {
            double s = 0;
            for (int sl = 0; sl < slices; ++sl) s += ho[(size_t)sl * slab + i];
            if (i < (size_t)H * H) err = std::max(err, std::fabs(s - ref[i])); else berr = std::max(berr, std::fabs(s - ref[i]));
        }
        printf("%s  small problem (%zu rows, %d slices): largest error %.3e of the largest product, bias %.3e of the largest bias\n",
               which ? "k_gemm_tn_ws" : "k_gemm_tn_w ", small, slices, err / rmax, berr / bmax);
    }
    // timing on the large problem (contents: whatever the small problem left, repeated zeros beyond: timing only)
    CK(hipMemset(X + small * H, 0, (big - small) * H * 4)); CK(hipMemset(Y + small * H, 0, (big - small) * H * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        TnProblems pr = {};
        pr.X[0] = X; pr.Y[0] = Y; pr.rows[0] = big; pr.nslices[0] = 256; pr.out[0] = out; pr.out_stride[0] = slab;
        launch(which, H, pr, 256);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch(which, H, pr, 256);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("%s  %zu rows: %.3f ms  %.1f TFLOP/s (fp32 product count)  %.2f TB/s of operand rows\n", which ? "k_gemm_tn_ws" : "k_gemm_tn_w ", big, ms,
               2.0 * big * H * H / ms / 1e9, 2.0 * big * H * 4 / ms / 1e9);
    }
    return 0;
}
