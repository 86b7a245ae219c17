#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; sm[threadIdx.x] = p[threadIdx.x]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = sm[1] + 1.f; }
__global__ void k_big(float* p, int n) { for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f; }
struct Big { int a[64]; };
__global__ void k_args(float* p, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.a[3]; }
int main() {
    float* d; hipMalloc(&d, 256 << 20); hipMemset(d, 0, 256 << 20);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 2000;
    auto run = [&](const char* name, auto f) {
        for (int i = 0; i < 50; ++i) f(i);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(a, s);
        for (int i = 0; i < N; ++i) f(i);
        hipEventRecord(b, s);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s gpu %.2f us/launch   host enqueue %.2f us/launch\n", name, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    };
    Big bg{}; bg.a[3] = 1;
    run("small back-to-back", [&](int) { k_small<<<1, 64, 0, s>>>(d); });
    run("small + hipGetLastError", [&](int) { k_small<<<1, 64, 0, s>>>(d); (void)hipGetLastError(); });
    run("256 WG x 512 thr, 144 KB LDS", [&](int) { k_lds<<<256, 512, 144 * 1024, s>>>(d); });
    run("alternate small / 144 KB LDS", [&](int i) { if (i & 1) k_lds<<<256, 512, 144 * 1024, s>>>(d); else k_small<<<1, 64, 0, s>>>(d); });
    run("256-byte kernarg", [&](int) { k_args<<<1, 64, 0, s>>>(d, bg); });
    run("64 MB read+write (dirty L2)", [&](int) { k_big<<<4096, 256, 0, s>>>(d, 16 << 20); });
    run("64 MB rw then small, per pair", [&](int i) { if (i & 1) k_small<<<1, 64, 0, s>>>(d); else k_big<<<4096, 256, 0, s>>>(d, 16 << 20); });
    run("4 MB read+write", [&](int) { k_big<<<1024, 256, 0, s>>>(d, 1 << 20); });
    return 0;
}
