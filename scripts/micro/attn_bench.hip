// Microbenchmark + check of the attention kernels (attention.hip), outside the library: one variant per process
// (launch_attn reads its switches once). Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-inline-asm -Iarchi_amd/csrc scripts/micro/attn_bench.hip -o scripts/micro/attn_bench
// Run:   AK_ATTN_STREAM=3 AK_ATTN_PIPE=1 scripts/micro/attn_bench <hd> <heads> <B> <S> <maskmode> [reps]
//   maskmode 0 = no padding, 1 = right-padded random lengths, 2 = left padding + holes, 3 = spiked scores (range check)
// Prints the kernel time (HIP events, average over reps after warm-up) and the error against a float32 reference kernel.
#define AK_DBG_KERNELS 1
#include "../../archi_amd/csrc/attention.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
namespace ak {
void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); }
}
using namespace ak;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one wave per (b, h, query): float32 softmax(q k^T + mask) v on the bf16 inputs (q pre-scaled for 2^x)
__global__ void k_ref(const uint16_t *q, const uint16_t *k, const uint16_t *vt, const int *mask, float *out, int B, int S, int H, int heads) {
    const int hd = H / heads;
    const int64_t row = blockIdx.x;
    const int qi = row % S, h = (row / S) % heads, b = row / ((int64_t)S * heads);
    const int lane = threadIdx.x;
    __shared__ float p[512];
    float mx = -INFINITY;
    for (int j = lane; j < S; j += 64) {
        float s = 0.f;
        for (int d = 0; d < hd; d++) s += bf16_to_f32(q[((int64_t)b * S + qi) * H + h * hd + d]) * bf16_to_f32(k[((int64_t)b * S + j) * H + h * hd + d]);
        if (!mask[b * S + j]) s = -INFINITY;
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < S; j += 64) { const float e = mx > -INFINITY ? exp2f(p[j] - mx) : 0.f; p[j] = e; sum += e; }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < S; j++) acc += p[j] * bf16_to_f32(vt[((int64_t)b * H + h * hd + d) * S + (j & ~31) + vt_pos(j & 31)]);
        out[((int64_t)b * S + qi) * H + h * hd + d] = sum > 0.f ? acc / sum : 0.f;
    }
}

int main(int argc, char **argv) {
    const int hd = argc > 1 ? atoi(argv[1]) : 64, heads = argc > 2 ? atoi(argv[2]) : 12, B = argc > 3 ? atoi(argv[3]) : 128;
    const int S = argc > 4 ? atoi(argv[4]) : 512, mode = argc > 5 ? atoi(argv[5]) : 0, reps = argc > 6 ? atoi(argv[6]) : 20;
    const int H = hd * heads;
    const int64_t T = (int64_t)B * S;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hq(T * H), hk(T * H), hvt((T + 512) * H);
    std::vector<int> hmask(T, 1);
    const float qs = 1.4426950408889634f / sqrtf((float)hd) * 2.5f;       // sharper than unit-variance scores
    for (auto &x : hq) x = f32_to_bf16(nd(rng) * qs);
    for (auto &x : hk) x = f32_to_bf16(nd(rng));
    for (auto &x : hvt) x = f32_to_bf16(nd(rng));
    if (mode == 3) {          // a few huge key rows far into the sequence: the running reference must move (range check path)
        for (int b = 0; b < B; b++)
            for (int t : {S / 2 + 3, S - 5})
                for (int c = 0; c < H; c++) hk[((int64_t)b * S + t) * H + c] = f32_to_bf16(nd(rng) * 40.f);
    }
    if (mode == 1) for (int b = 0; b < B; b++) { const int len = 1 + rng() % S; for (int j = len; j < S; j++) hmask[b * S + j] = 0; }
    if (mode == 2) for (int b = 0; b < B; b++) {
        const int lp = rng() % (S / 2);
        for (int j = 0; j < lp; j++) hmask[b * S + j] = 0;
        for (int j = lp; j < S; j++) if (rng() % 5 == 0) hmask[b * S + j] = 0;
        if (b % 3 == 0) for (int j = S / 2; j < S / 2 + 70 && j < S; j++) hmask[b * S + j] = 0;
        hmask[b * S + S - 1] = 1;
    }
    uint16_t *q, *k, *vt, *ctx; int *mask; float *maskf, *ref;
    CK(hipMalloc(&q, T * H * 2)); CK(hipMalloc(&k, T * H * 2)); CK(hipMalloc(&vt, (T + 512) * H * 2)); CK(hipMalloc(&ctx, T * H * 2));
    CK(hipMalloc(&mask, T * 4)); CK(hipMalloc(&maskf, T * 4 + (T / 32 + 1) * 4)); CK(hipMalloc(&ref, T * H * 4));
    CK(hipMemcpy(q, hq.data(), T * H * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), T * H * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hvt.data(), (T + 512) * H * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(mask, hmask.data(), T * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ctx, 0, T * H * 2));
    if (launch_attn_prepare(mask, B, S, maskf, (uint32_t *)(maskf + T), 0)) return 1;
    AttnArgs a{q, k, vt, mask, ctx, B, S, H, heads, maskf, (const uint32_t *)(maskf + T), 0, 0};
    for (int i = 0; i < 3; i++) if (launch_attn(a, 0)) return 1;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch_attn(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    k_ref<<<(unsigned)(T * heads), 64>>>(q, k, vt, mask, ref, B, S, H, heads);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> hc(T * H); std::vector<float> hr(T * H);
    CK(hipMemcpy(hc.data(), ctx, T * H * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), ref, T * H * 4, hipMemcpyDeviceToHost));
    double maxe = 0, sume = 0; int64_t bad = 0, nan = 0;
    for (int64_t i = 0; i < T * H; i++) {
        const float c = bf16_to_f32(hc[i]);
        if (c != c) { nan++; continue; }
        const double e = fabs((double)c - hr[i]);
        maxe = e > maxe ? e : maxe; sume += e;
        if (e > 0.02 + 0.01 * fabs(hr[i])) bad++;
    }
    const double flop = 4.0 * B * heads * (double)S * S * hd;
    printf("hd %d heads %d B %d S %d mask %d | STREAM=%s PIPE=%s NW=%s | %.1f us  %.0f TF | max err %.4f mean err %.6f bad %lld nan %lld\n", hd, heads, B, S, mode,
           getenv("AK_ATTN_STREAM") ? getenv("AK_ATTN_STREAM") : "-", getenv("AK_ATTN_PIPE") ? getenv("AK_ATTN_PIPE") : "-",
           getenv("AK_ATTN_NW") ? getenv("AK_ATTN_NW") : "-", ms * 1000 / reps, flop / (ms / reps * 1e-3) / 1e12, maxe, sume / (T * H), (long long)bad, (long long)nan);
    return (bad || nan) ? 2 : 0;
}
