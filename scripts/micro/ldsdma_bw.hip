// Microbenchmark: L2 -> LDS throughput of global_load_lds_dwordx4 per CU, for the weight-streaming pattern of the fused layer
// kernels (every CU walks the same 2.3 MB in 48 KB chunks). hipcc --offload-arch=gfx950 -O3 ldsdma_bw.hip -o ldsdma_bw
//   mode bits: 1 = every workgroup starts at a different chunk (rotation), 2 = only waves 4-7 issue (12 pieces each, burst),
//              4 = keep one chunk in flight across the barrier (counted vmcnt), 8 = nt (aux) on the loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int SLOT = 48 * 1024, NCH = 48, NSLOT = 3;
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const char *w, int iters, long long *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)smem;
    const uint32_t voff = lane * 16;
    constexpr bool ROT = MODE & 1, HALF = MODE & 2, DEEP = MODE & 4, NT = MODE & 8;
    constexpr int PPW = HALF ? 12 : 6;
    const bool issuer = HALF ? wave >= 4 : true;
    const int w0 = HALF ? (wave - 4) * 12 : wave * 6;
    const int rot = ROT ? (blockIdx.x * 7) % NCH : 0;
    auto stage = [&](int it) {
        const int ch = (it + rot) % NCH;
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const char *base = w + (size_t)ch * SLOT + (w0 + i) * 1024;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (it % NSLOT) * SLOT + (w0 + i) * 1024);
            if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    long long t0 = 0;
    for (int rep = 0; rep < 2; rep++) {      // rep 0 warms L2
        __syncthreads();
        t0 = (long long)__builtin_readcyclecounter();
        if (issuer) stage(0);
        if (DEEP && issuer) stage(1);
        for (int it = 0; it < iters; it++) {
            if (DEEP) { if (it + 1 < iters) wait_vm<PPW>(); else wait_vm<0>(); } else wait_vm<0>();
            __syncthreads();
            const int nx = it + (DEEP ? 2 : 1);
            if (issuer && nx < iters) stage(nx);
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 1) out[gridDim.x + blockIdx.x] = *(volatile int *)(smem + 4 * (iters & 15));
}
template <int MODE> void run(const char *w, long long *out, int grid, int iters) {
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<grid, 512, NSLOT * SLOT>>>(w, iters, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<MODE><<<grid, 512, NSLOT * SLOT>>>(w, iters, out); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid); CK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    const double cyc = s / grid / iters;
    printf("mode %2d grid %3d: %.0f cycles per 48 KB chunk = %.1f B/clk/CU; launch %.3f ms (2 x %d chunks) -> %.2f TB/s chip\n", MODE, grid, cyc, SLOT / cyc,
           ms, iters, 2.0 * iters * SLOT * grid / (ms * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 480;
    char *w; long long *out;
    CK(hipMalloc((void **)&w, (size_t)NCH * SLOT)); CK(hipMemset(w, 1, (size_t)NCH * SLOT)); CK(hipMalloc((void **)&out, 4096 * 8));
    run<0>(w, out, grid, iters); run<1>(w, out, grid, iters); run<2>(w, out, grid, iters); run<3>(w, out, grid, iters);
    run<4>(w, out, grid, iters); run<5>(w, out, grid, iters); run<6>(w, out, grid, iters); run<7>(w, out, grid, iters);
    run<8>(w, out, grid, iters); run<12>(w, out, grid, iters); run<13>(w, out, grid, iters);
    return 0;
}
