// Microbenchmark: issue cost of the instructions the attention softmax is made of, alone and beside another wave on the same
// SIMD (v_exp_f32 / v_add_f32 / v_fma_f32 / v_pk_add_f32 / v_cvt_pk_bf16_f32 / v_exp_f16 / packed f16 ops / MFMA).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_rates.hip -o scripts/micro/issue_rates
// A workgroup has 8 waves; waves w and w + 4 share a SIMD. Each wave runs ITERS x 32 independent instructions of its role and
// reports shader cycles per instruction. Role 0 = exit at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Roles { int r[8]; };
enum { R_EXIT = 0, R_EXP, R_FMA, R_ADD, R_PKADD, R_CVTPK, R_EXPF16, R_PKFMAF16, R_MFMA32, R_MFMA16, R_MOV, R_LDEXP, R_NROLES };
static const char *names[] = {"-", "v_exp_f32", "v_fma_f32", "v_add_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_exp_f16", "v_pk_fma_f16",
                              "mfma_32x32x16_bf16", "mfma_16x16x32_bf16", "v_mov_b32", "v_ldexp_f32"};

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
__global__ __launch_bounds__(512) void k(Roles ro, int iters, long long *out) {
    const int wave = threadIdx.x >> 6;
    const int role = ro.r[wave];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    f32x16 acc[2] = {};
    f32x4 acc4[4] = {};
    bf16x8 fa = {}, fb = {};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (role == R_EXP) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                REP8(S)
#undef S
            }
    } else if (role == R_FMA) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
                REP8(S)
#undef S
            }
    } else if (role == R_ADD) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
                REP8(S)
#undef S
            }
    } else if (role == R_MOV) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_mov_b32 %0, %0" : "+v"(v[i]));
                REP8(S)
#undef S
            }
    } else if (role == R_LDEXP) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[i]));
                REP8(S)
#undef S
            }
    } else if (role == R_PKADD) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p[8];
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = f2{v[i], v[i]};
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
                REP8(S)
#undef S
            }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = p[i][0] + p[i][1];
    } else if (role == R_CVTPK) {
        uint32_t u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = __float_as_uint(v[i]);
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(u[i]));
                REP8(S)
#undef S
            }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __uint_as_float(u[i]);
    } else if (role == R_EXPF16) {
        uint32_t u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = threadIdx.x + i;
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_exp_f16 %0, %0" : "+v"(u[i]));
                REP8(S)
#undef S
            }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __uint_as_float(u[i]);
    } else if (role == R_PKFMAF16) {
        uint32_t u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = threadIdx.x + i;
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 4; j++) {
#define S(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(u[i]));
                REP8(S)
#undef S
            }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __uint_as_float(u[i]);
    } else if (role == R_MFMA32) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 16; j++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[1], 0, 0, 0);
            }
    } else if (role == R_MFMA16) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4[q], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    s += acc[0][0] + acc[1][3] + acc4[0][0] + acc4[1][1] + acc4[2][2] + acc4[3][3];
    if (s == 12345.678f) out[1000000] = 1;
    if ((threadIdx.x & 63) == 0 && role != R_EXIT) out[blockIdx.x * 8 + wave] = t1 - t0;
}

static void run(const char *label, std::initializer_list<int> roles, int iters, long long *dout) {
    Roles ro{};
    int i = 0;
    for (int r : roles) ro.r[i++] = r;
    CK(hipMemset(dout, 0, 256 * 8 * sizeof(long long)));
    k<<<256, 512>>>(ro, iters, dout);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(256 * 8);
    CK(hipMemcpy(h.data(), dout, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    printf("%-34s", label);
    for (int w = 0; w < 8; w++) {
        if (!ro.r[w]) continue;
        double sum = 0;
        for (int b = 0; b < 256; b++) sum += (double)h[b * 8 + w];
        printf("  w%d %-18s %6.2f cyc/instr", w, names[ro.r[w]], sum / 256 / (iters * 32.0));
    }
    printf("\n");
}

int main() {
    long long *dout; CK(hipMalloc(&dout, (1000008) * sizeof(long long)));
    const int it = 2000;
    for (int pass = 0; pass < 2; pass++) {
        for (int r = 1; r < R_NROLES; r++) run("alone", {r, 0, 0, 0, 0, 0, 0, 0}, it, dout);
        run("same SIMD: exp + exp", {R_EXP, 0, 0, 0, R_EXP, 0, 0, 0}, it, dout);
        run("same SIMD: exp + fma", {R_EXP, 0, 0, 0, R_FMA, 0, 0, 0}, it, dout);
        run("same SIMD: exp + add", {R_EXP, 0, 0, 0, R_ADD, 0, 0, 0}, it, dout);
        run("same SIMD: fma + fma", {R_FMA, 0, 0, 0, R_FMA, 0, 0, 0}, it, dout);
        run("same SIMD: exp + mfma32", {R_EXP, 0, 0, 0, R_MFMA32, 0, 0, 0}, it, dout);
        run("same SIMD: fma + mfma32", {R_FMA, 0, 0, 0, R_MFMA32, 0, 0, 0}, it, dout);
        run("same SIMD: exp + mfma16", {R_EXP, 0, 0, 0, R_MFMA16, 0, 0, 0}, it, dout);
        run("same SIMD: mfma32 + mfma32", {R_MFMA32, 0, 0, 0, R_MFMA32, 0, 0, 0}, it, dout);
        run("same SIMD: expf16 + fma", {R_EXPF16, 0, 0, 0, R_FMA, 0, 0, 0}, it, dout);
        run("same SIMD: pkfmaf16 + exp", {R_PKFMAF16, 0, 0, 0, R_EXP, 0, 0, 0}, it, dout);
        run("all SIMDs exp (4 waves)", {R_EXP, R_EXP, R_EXP, R_EXP, 0, 0, 0, 0}, it, dout);
        run("all SIMDs exp x2 (8 waves)", {R_EXP, R_EXP, R_EXP, R_EXP, R_EXP, R_EXP, R_EXP, R_EXP}, it, dout);
        run("all SIMDs exp + mfma32", {R_EXP, R_EXP, R_EXP, R_EXP, R_MFMA32, R_MFMA32, R_MFMA32, R_MFMA32}, it, dout);
    }
    return 0;
}
