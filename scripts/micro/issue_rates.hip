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
enum { R_EXIT = 0, R_EXP, R_FMA, R_ADD, R_PKADD, R_CVTPK, R_EXPF16, R_PKFMAF16, R_MFMA32, R_MFMA16, R_MOV, R_LDEXP, R_MIX_EXP4, R_MIX_FMA5, R_MIX_EXP2, R_MIX16_EXP2, R_MIX_DS, R_NROLES };
static const char *names[] = {"-", "v_exp_f32", "v_fma_f32", "v_add_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_exp_f16", "v_pk_fma_f16",
                              "mfma_32x32x16_bf16", "mfma_16x16x32_bf16", "v_mov_b32", "v_ldexp_f32", "[mfma32 + 4 exp]/5", "[mfma32 + 5 fma]/6", "[mfma32 + 2 exp]/3", "[mfma16 + 2 exp]/3", "[mfma32 + 2 ds_read_b128 + 3 fma]/6"};

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
    } else if (role == R_MIX_EXP4) {       // per MFMA: 4 independent exps of the same wave (32 instructions = 6.4 groups per j-iteration... counted as instructions)
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 32; j += 5) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(fa), "v"(fb));
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            }
    } else if (role == R_MIX_FMA5) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 32; j += 6) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(fa), "v"(fb));
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
            }
    } else if (role == R_MIX_EXP2) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 32; j += 3) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(fa), "v"(fb));
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[j & 3]), "+v"(v[4 + (j & 3)]));
            }
    } else if (role == R_MIX16_EXP2) {
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 32; j += 3) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[j & 3]) : "v"(fa), "v"(fb));
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[j & 3]), "+v"(v[4 + (j & 3)]));
            }
    } else if (role == R_MIX_DS) {
        __shared__ float4 sm[512];
        float4 d0, d1;
        for (int it = 0; it < iters; it++)
            for (int j = 0; j < 32; j += 6) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(fa), "v"(fb));
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(d0), "=v"(d1) : "v"((uint32_t)(threadIdx.x * 16)), "v"(v[0]), "v"(v[1]), "v"(v[2]));
                v[3] += d0.x + d1.y;
            }
        if (v[3] == 1.2345f) sm[threadIdx.x] = d0;
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
        { const int r = ro.r[w]; const double per = r == R_MIX_EXP4 ? 7.0 : r == R_MIX_FMA5 ? 6.0 : (r == R_MIX_EXP2 || r == R_MIX16_EXP2) ? 11.0 : r == R_MIX_DS ? 6.0 : 32.0;
          printf("  w%d %-18s %6.2f cyc/%s", w, names[r], sum / 256 / (iters * per), per == 32.0 ? "instr" : "group"); }
    }
    printf("\n");
}

int main() {
    long long *dout; CK(hipMalloc(&dout, (1000008) * sizeof(long long)));
    const int it = 2000;
    for (int pass = 0; pass < 2; pass++) {
        for (int r = 1; r < R_MIX_EXP4; r++) run("alone", {r, 0, 0, 0, 0, 0, 0, 0}, it, dout);
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
        for (int r = R_MIX_EXP4; r < R_NROLES; r++) run("alone (one stream, MFMA + VALU)", {r, 0, 0, 0, 0, 0, 0, 0}, it, dout);
        for (int r = R_MIX_EXP4; r < R_NROLES; r++) run("two such waves on one SIMD", {r, 0, 0, 0, r, 0, 0, 0}, it, dout);
        run("four waves [mfma32 + 4 exp] on one SIMD: n/a (8-wave workgroup has two per SIMD)", {R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4, R_MIX_EXP4}, it, dout);
    }
    return 0;
}
