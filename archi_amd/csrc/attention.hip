// attention.hip -- bidirectional multi-head self-attention of the BERT encoder (a1/a2):
//   ctx = softmax(q k^T / sqrt(hd) + mask) v      per (sequence, head), hd = 32 or 64, S <= 512
// q is pre-scaled by log2(e)/sqrt(hd) (so the softmax runs on v_exp_f32 = 2^x directly) and v arrives
// transposed ([B][H][S]) from the QKV GEMM epilogue (gemm.hip), with the keys of every group of 16 stored in the
// order [0-3, 8-11, 4-7, 12-15] (vt_pos): the 16 bytes a lane feeds to one P.V MFMA are then contiguous.
//
// One wave owns 32 queries. The score tile is computed SWAPPED (A = keys, B = queries), so a lane
// owns one query column and 16 keys per 32x32 MFMA tile: the softmax max/sum are lane-local plus
// one exchange with lane^32. P feeds the P.V MFMA straight from registers (K-order of the two
// operands is chosen to match the accumulator layout, no LDS round trip). Keys are processed in
// chunks of 128 with an online-softmax rescale, so S = 512 fits the register file.
#include <atomic>
#include <vector>
#include <cstdio>
#include "mfma_tile.h"
#include "encoder_kernels.h"
#include "switches.h"
#include "attn_d.h"

namespace ak {
using namespace mt;

// (An inline-asm v_max3_f32 on MFMA outputs, tried to avoid hipcc's canonicalising v_max, read the accumulators BEFORE the MFMA
// had written them: the hazard recogniser does not cover inline-asm operands. Results differed from run to run.)
template <int HD, int NW, int CB = 4>   // CB: 32-key blocks per online-softmax chunk
__global__ __launch_bounds__(NW * 64, (HD == 32 && NW <= 8 && CB == 2) ? 6 : ((HD == 32 || NW == 16) ? 4 : 2)) void k_attn(AttnArgs a) {
    constexpr int NT = NW * 64;
    constexpr int DB = HD / 32, KSTEPS = HD / 16;
    constexpr int KSTRIDE = HD * 2 + 16;             // padded K row (bytes): conflict-free b128 reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, H = a.H;
    const int VSTRIDE = S * 2 + 16;                  // padded V^T row (bytes)
    char *sK = smem;
    char *sV = sK + S * KSTRIDE;
    float *sM = (float *)(sV + HD * VSTRIDE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;

    // this lane's query fragments first: the loads fly under the staging below
    const int r = lane & 31, kh = lane >> 5;
    int qrow = q0 + r;
    if (qrow >= S) qrow = S - 1;
    uint4 qf[KSTEPS];
#pragma unroll
    for (int st = 0; st < KSTEPS; st++)
        qf[st] = *(const uint4 *)(a.q + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)qrow * a.qk_ld + st * 16 + kh * 8);

    // ---- stage K [S][HD], V^T [HD][S], mask for this (b, h): the 16-B chunks of both arrays are one index space and a
    // thread issues up to 8 loads before the first LDS write (a load -> wait -> write loop per chunk serialised 8 memory
    // latencies per workgroup: 128 KB per workgroup at hd = 64, S = 512, nothing else resident on the CU to cover them)
    {
        const uint16_t *kg = a.k + ((int64_t)b * S) * H + (int64_t)h * a.qk_hs;
        const uint16_t *vg = a.vt + ((int64_t)b * H + h * HD) * S;
        constexpr int KC = HD / 8;                   // 16-B chunks per K row
        const int VC = S / 8;
        const int totk = S * KC, totv = HD * VC;
        int mraw = tid < S ? a.mask[b * S + tid] : 0;     // consumed after the K / V^T writes (the pin below keeps hipcc from waiting here)
        constexpr int UN = 4;
        const int vstep_d = NT / VC, vstep_c = NT - vstep_d * VC;   // V^T chunk index -> (row, chunk) without a division per chunk
        int vd = tid / VC, vc = tid - vd * VC;
        for (int base = 0; base < totk || base < totv; base += UN * NT) {
            uint4 tk[UN], tv[UN];
            int vrow[UN], vcol[UN];
#pragma unroll
            for (int j = 0; j < UN; j++) {
                const int i = base + j * NT + tid;
                tk[j] = uint4{0, 0, 0, 0};
                if (i < totk) tk[j] = *(const uint4 *)(kg + (int64_t)(i / KC) * a.qk_ld + (i % KC) * 8);
            }
#pragma unroll
            for (int j = 0; j < UN; j++) {
                const int i = base + j * NT + tid;
                vrow[j] = vd; vcol[j] = vc;
                tv[j] = uint4{0, 0, 0, 0};
                if (i < totv) tv[j] = *(const uint4 *)(vg + (int64_t)vd * S + vc * 8);
                vd += vstep_d; vc += vstep_c;
                if (vc >= VC) { vc -= VC; vd++; }
            }
#pragma unroll
            for (int j = 0; j < UN; j++) {
                const int i = base + j * NT + tid;
                if (i < totk) *(uint4 *)(sK + (i / KC) * KSTRIDE + (i % KC) * 16) = tk[j];
            }
#pragma unroll
            for (int j = 0; j < UN; j++) {
                const int i = base + j * NT + tid;
                if (i < totv) *(uint4 *)(sV + vrow[j] * VSTRIDE + vcol[j] * 16) = tv[j];
            }
        }
        asm volatile("" : "+v"(mraw));
        if (tid < S) sM[tid] = mraw ? 0.f : -__builtin_inff();
        for (int i = tid + NT; i < S; i += NT) sM[i] = a.mask[b * S + i] ? 0.f : -__builtin_inff();
    }
    __syncthreads();
    if (q0 >= S) return;

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    float m = -__builtin_inff();
    f32x2 l2 = {0.f, 0.f};                             // row sum as two partial sums (packed adds)

    for (int kc0 = 0; kc0 < S; kc0 += 32 * CB) {
        const int nblk = (S - kc0) >= 32 * CB ? CB : (S - kc0) / 32;
        f32x16 sc[CB];
#pragma unroll
        for (int blk = 0; blk < CB; blk++) {
            if (blk >= nblk) {       // ragged tail chunk only: keys past S contribute nothing
#pragma unroll
                for (int e = 0; e < 16; e++) sc[blk][e] = -__builtin_inff();
            } else {
                // the additive mask (0 / -inf per key = per accumulator row) is the MFMA's initial accumulator:
                // no zero fill and no separate add
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 mk = *(const float4 *)&sM[kc0 + blk * 32 + 8 * g + 4 * kh];
                    acc[4 * g + 0] = mk.x; acc[4 * g + 1] = mk.y; acc[4 * g + 2] = mk.z; acc[4 * g + 3] = mk.w;
                }
                const char *kr = sK + (kc0 + blk * 32 + r) * KSTRIDE + kh * 16;
#pragma unroll
                for (int st = 0; st < KSTEPS; st++) acc = mfma_bf16(*(const uint4 *)(kr + st * 32), qf[st], acc);
                sc[blk] = acc;
            }
        }
        // LAZY running maximum: the scores are taken relative to the maximum m seen so far, and m is only raised (O, l and
        // this chunk rescaled) when some query's chunk maximum exceeds it by more than 2^8 -- a wave-uniform branch, rarely
        // taken after the first chunk; p <= 2^8 is harmless in the fp32 sums and the bf16 P operand. The subtraction comes
        // first: its results need no canonicalising v_max in front of the maximum (MFMA outputs do), and it packs two scores
        // per instruction (v_pk_add_f32), like the row sum below.
        const bool fresh = m == -__builtin_inff();   // no unmasked key seen yet
        const float m_use = fresh ? 0.f : m;
        const f32x2 mm = {m_use, m_use};
        float mx = -__builtin_inff();
#pragma unroll
        for (int blk = 0; blk < CB; blk++)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x = f32x2{sc[blk][e], sc[blk][e + 1]} - mm;
                sc[blk][e] = x[0]; sc[blk][e + 1] = x[1];
                mx = fmaxf(mx, fmaxf(x[0], x[1]));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(fresh ? mx > -__builtin_inff() : mx > 8.f)) {
            const float delta = fresh ? (mx > -__builtin_inff() ? mx : 0.f) : fmaxf(mx, 0.f);
            const float alpha = fresh ? 0.f : __builtin_amdgcn_exp2f(-delta);
            if (!fresh || mx > -__builtin_inff()) m = m_use + delta;
            const f32x2 dd = {delta, delta};
            l2 *= alpha;
#pragma unroll
            for (int d = 0; d < DB; d++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[d][e] *= alpha;
#pragma unroll
            for (int blk = 0; blk < CB; blk++)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 x = f32x2{sc[blk][e], sc[blk][e + 1]} - dd;
                    sc[blk][e] = x[0]; sc[blk][e + 1] = x[1];
                }
        }
#pragma unroll
        for (int blk = 0; blk < CB; blk++)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 pv = {__builtin_amdgcn_exp2f(sc[blk][e]), __builtin_amdgcn_exp2f(sc[blk][e + 1])};
                sc[blk][e] = pv[0]; sc[blk][e + 1] = pv[1];
                l2 += pv;
            }
#pragma unroll
        for (int blk = 0; blk < CB; blk++) {
            if (blk >= nblk) break;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                uint4 pb = {pack_bf16x2(sc[blk][8 * s2 + 0], sc[blk][8 * s2 + 1]), pack_bf16x2(sc[blk][8 * s2 + 2], sc[blk][8 * s2 + 3]),
                            pack_bf16x2(sc[blk][8 * s2 + 4], sc[blk][8 * s2 + 5]), pack_bf16x2(sc[blk][8 * s2 + 6], sc[blk][8 * s2 + 7])};
#pragma unroll
                for (int d = 0; d < DB; d++) {
                    const uint4 va = *(const uint4 *)(sV + (d * 32 + r) * VSTRIDE + (kc0 + blk * 32 + 16 * s2 + 8 * kh) * 2);
                    o[d] = mfma_bf16(va, pb, o[d]);
                }
            }
        }
    }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    {
        const int orow = q0 + r < S ? q0 + r : S - 1;
        store_ctx_rows<DB>(o, inv, a.ctx + ((int64_t)b * S + orow) * H + h * HD, kh, q0 + r < S);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streamed variant (launched for hd = 64; AK_ATTN_STREAM=0 / 1 selects per run for A/B).
// Measured on k_attn (AK_ATTN_ABLATE, rocprofv3 kernel durations, bge-base 128 x 512 / MiniLM 256 x 256): 201 / 65 us,
// of which exp 16 / 10, the rest of the softmax 7 / 2, P.V 28 / 5, Q.K^T 45 / 9 and the STAGING 70 / 13 -- every
// workgroup loaded its K / V^T (128 KB at hd = 64, S = 512: one workgroup per CU), waited, computed, stored: all 256 CUs
// load together, then all compute together. Here a workgroup is persistent over (sequence, head, query block) items
// and K / V^T / the additive mask arrive by LDS-DMA in key tiles of 256 / 128 / 64 / 32 keys through a two-slot ring:
// the tile after the current one is in flight while the current one is computed (one barrier per tile), the next
// item's query fragments are loaded during the current item's last tile. LDS rows are unpadded and XOR-swizzled on the
// SOURCE address so that every 16-lane group of a ds_read_b128 covers all 64 banks. Softmax: 32-key blocks with a LAZY
// running maximum -- the row maximum is only raised (and O, l rescaled) when some query's block maximum exceeds it by
// more than 2^8, a wave-uniform branch that is rarely taken after the first block; p <= 2^8 is harmless in the fp32
// sums and the bf16 P operand. 32-key blocks without a real key (padding) are skipped.
__global__ __launch_bounds__(512) void k_attn_prepare(const int *__restrict__ mask, int S, float *__restrict__ maskf,
                                                      uint32_t *__restrict__ blkmask) {
    __shared__ uint32_t bits;
    const int b = blockIdx.x, t = threadIdx.x;          // one thread per key (S <= 512): one load latency per sequence
    if (t == 0) bits = 0;
    __syncthreads();
    const int mv = t < S ? mask[b * S + t] : 0;
    if (t < S) maskf[b * S + t] = mv ? 0.f : -__builtin_inff();
    const uint64_t bal = __ballot(mv != 0);
    // low half: 32-key blocks with a real key; high half (bit 16 + block): blocks whose 32 keys are ALL real -- their additive mask
    // is all zero, so the score MFMA starts from a zero accumulator instead of four LDS reads of the mask (S <= 512: 16 blocks)
    const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
    if ((t & 63) == 0 && bal)
        atomicOr(&bits, ((lo ? 1u : 0u) | (hi ? 2u : 0u) | (lo == 0xffffffffu ? 0x10000u : 0u) | (hi == 0xffffffffu ? 0x20000u : 0u)) << (t >> 5));
    __syncthreads();
    if (t == 0) blkmask[b] = bits;
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_attn_s(AttnArgs a, int nitems, int ktm) {
    constexpr int DB = HD / 32, KSTEPS = HD / 16, KROW = HD * 2, CRK = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, H = a.H, heads = a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    // ring slot: K tile [kt][HD] | V^T tile [HD][kt] | additive mask [kt] f32, sized for the largest tile of this S
    const int slot_k = ktm * KROW, slot_v = HD * ktm * 2, slot = slot_k + slot_v + ktm * 4;
    const uint32_t lds0 = lds_addr(smem);
    const int nqb = (S + NW * 32 - 1) / (NW * 32);
    const int n256 = S >> 8, rem = S & 255;
    const int ntl = n256 + __popc(rem >> 5);          // key tiles per item: 256-key tiles, then 128 / 64 / 32
    auto tile_at = [&](int j, int &k0, int &kt) {
        k0 = (j < n256 ? j : n256) << 8; kt = 256;
        if (j >= n256) {
            int jj = j - n256;
            for (int bit = 128; bit >= 32; bit >>= 1)
                if (rem & bit) {
                    if (jj == 0) { kt = bit; break; }
                    jj--; k0 += bit;
                }
        }
    };
    // lane constants of the swizzles: K rows hold CRK chunks (chunk ^= (row>>1)&7 at 128 B, (row>>2)&3 at 64 B)
    const int kx = (kh ^ (HD == 64 ? (r >> 1) & 7 : (r >> 2) & 3)) << 4;

    auto issue = [&](int it, int j, int sl) {          // tile j of item `it` -> ring slot sl
        const int bh = it / nqb, b = bh / heads, h = bh - b * heads;
        int k0, kt; tile_at(j, k0, kt);
        const uint32_t sb = lds0 + sl * slot;
        const int nkp = (kt * KROW) >> 10;             // 1 KiB pieces of the K tile; the V^T tile has as many
        const char *kg = (const char *)(a.k + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)k0 * a.qk_ld);
        const char *vg = (const char *)(a.vt + ((int64_t)b * H + h * HD) * S + k0);
        const int lcr = 31 - __clz(kt >> 3);           // log2(16-B chunks per V^T tile row)
        const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
        for (int p = wave; p < 2 * nkp + 1; p += NW) {
            if (p < nkp) {
                const int g = p * 64 + lane, row = g / CRK, c = g % CRK;
                const int swz = HD == 64 ? (row >> 1) & 7 : (row >> 2) & 3;
                glds16(kg + (int64_t)row * a.qk_ld * 2 + ((c ^ swz) << 4), sb + p * 1024);
            } else if (p < 2 * nkp) {
                const int g = (p - nkp) * 64 + lane, row = g >> lcr, c = g & ((1 << lcr) - 1);
                const int swz = (row >> vsh) & vmsk;
                glds16(vg + (int64_t)row * S * 2 + ((c ^ swz) << 4), sb + slot_k + (p - nkp) * 1024);
            } else if (lane * 4 < kt) {
                glds16(a.maskf + (int64_t)b * S + k0 + lane * 4, sb + slot_k + slot_v);
            }
        }
    };
    auto load_q = [&](int it, uint4 (&qv)[KSTEPS]) {
        const int bh = it / nqb, qb = it - bh * nqb, b = bh / heads, h = bh - b * heads;
        int qrow = qb * (NW * 32) + wave * 32 + r;
        if (qrow >= S) qrow = S - 1;
#pragma unroll
        for (int st = 0; st < KSTEPS; st++)
            qv[st] = *(const uint4 *)(a.q + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)qrow * a.qk_ld + st * 16 + kh * 8);
    };

    int it = blockIdx.x, j = 0, sl = 0;
    uint4 qf[KSTEPS], qn[KSTEPS];
    f32x16 o[DB];
    float m = 0.f;
    f32x2 l2 = {0.f, 0.f};                             // row sum, two partial sums (packed adds)
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    issue(it, 0, 0);
    load_q(it, qf);
#pragma unroll
    for (int st = 0; st < KSTEPS; st++) qn[st] = qf[st];

    while (true) {
        // this item's key-block bitmap, fetched and made uniform BEFORE the wait below: hipcc issues a vector load for it and waits
        // vmcnt(0) where it is consumed -- after the issue of the next tile that would be a wait for the DMA just started
        const uint32_t flags_raw = __builtin_amdgcn_readfirstlane(a.blkmask[(it / nqb) / heads]);
        const uint32_t flags_all = flags_raw & 0xffffu, full_all = flags_raw >> 16;
        const int fb = flags_all ? __builtin_ctz(flags_all) : -1;
        int nit = it, nj = j + 1;
        if (nj == ntl) { nit = it + gridDim.x; nj = 0; }
        const bool has_next = nit < nitems;
        wait_vm<0>();                                   // this wave's pieces of tile (it, j) have landed
        __syncthreads();                                // ... everybody's have, and everybody is done with the other slot
        if (has_next) issue(nit, nj, sl ^ 1);
        const bool last = j == ntl - 1;
        if (last && has_next) load_q(nit, qn);

        const int bh = it / nqb, qb = it - bh * nqb, b = bh / heads, h = bh - b * heads;
        const int q0 = qb * (NW * 32) + wave * 32;
        if (q0 < S) {
            int k0, kt; tile_at(j, k0, kt);
            const char *sb = smem + sl * slot;
            const char *sV = sb + slot_k;
            const float *sM = (const float *)(sb + slot_k + slot_v);
            const int lcr = 31 - __clz(kt >> 3);
            const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
            const int vx = (kh ^ ((r >> vsh) & vmsk)) << 4;
            const uint32_t flags = flags_all >> (k0 >> 5), fullf = full_all >> (k0 >> 5);
            const char *krow = sb + r * KROW;
            const char *vrow = sV + r * (kt * 2);
            for (int blk = 0; blk < (kt >> 5); blk++) {
                if (!((flags >> blk) & 1)) continue;    // padding only: contributes exp2(-inf) = 0 to every sum
                // the additive mask (0 / -inf per key = per accumulator row) is the MFMA's initial accumulator; a block of 32 real
                // keys (wave-uniform flag) starts from the zero constant instead
                f32x16 acc;
                const char *kr = krow + blk * 32 * KROW;
                if ((fullf >> blk) & 1) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = mfma_bf16(*(const uint4 *)(kr + kx), qf[0], z);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float4 mk = *(const float4 *)&sM[blk * 32 + 8 * g + 4 * kh];
                        acc[4 * g + 0] = mk.x; acc[4 * g + 1] = mk.y; acc[4 * g + 2] = mk.z; acc[4 * g + 3] = mk.w;
                    }
                    acc = mfma_bf16(*(const uint4 *)(kr + kx), qf[0], acc);
                }
#pragma unroll
                for (int st = 1; st < KSTEPS; st++) acc = mfma_bf16(*(const uint4 *)(kr + ((st << 5) ^ kx)), qf[st], acc);
                // lazy running maximum, subtraction first (see k_attn); the item's first live block is the same for every query
                // (the padding mask is per key): a wave-uniform flag with scalar selects, as in k_attn_d
                const bool first = (k0 >> 5) + blk == fb;
                const f32x2 mm = {m, m};                // m = 0 until the first live block has set it
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 x = f32x2{acc[e], acc[e + 1]} - mm;
                    acc[e] = x[0]; acc[e + 1] = x[1];
                }
#if AK_DBG_KERNELS
                float mx = -__builtin_inff();         // (A/B reference: the fmaxf chain of rounds 2-4)
#pragma unroll
                for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(acc[e], acc[e + 1]));
#else
                float mx = max16(acc);
#endif
                mx = xhalf_max(mx);
                if (first || __any(mx > 8.f)) {
                    const float delta = first ? (mx > -__builtin_inff() ? mx : 0.f) : fmaxf(mx, 0.f);
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
                    m += delta;
                    const f32x2 dd = {delta, delta};
                    l2 *= alpha;
#pragma unroll
                    for (int d = 0; d < DB; d++)
#pragma unroll
                        for (int e = 0; e < 16; e++) o[d][e] *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 x = f32x2{acc[e], acc[e + 1]} - dd;
                        acc[e] = x[0]; acc[e + 1] = x[1];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 pv = {__builtin_amdgcn_exp2f(acc[e]), __builtin_amdgcn_exp2f(acc[e + 1])};
                    acc[e] = pv[0]; acc[e + 1] = pv[1];
                    l2 += pv;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const uint4 pb = {pack_bf16x2(acc[8 * s2 + 0], acc[8 * s2 + 1]), pack_bf16x2(acc[8 * s2 + 2], acc[8 * s2 + 3]),
                                      pack_bf16x2(acc[8 * s2 + 4], acc[8 * s2 + 5]), pack_bf16x2(acc[8 * s2 + 6], acc[8 * s2 + 7])};
#pragma unroll
                    for (int d = 0; d < DB; d++) {
                        const uint4 va = *(const uint4 *)(vrow + d * 32 * (kt * 2) + ((blk * 64 + s2 * 32) ^ vx));
                        o[d] = mfma_bf16(va, pb, o[d]);
                    }
                }
            }
            if (last) {
                float l = l2[0] + l2[1];
                l += __shfl_xor(l, 32);
                const float inv = l > 0.f ? 1.0f / l : 0.f;
                {
                    const int orow = q0 + r < S ? q0 + r : S - 1;
                    store_ctx_rows<DB>(o, inv, a.ctx + ((int64_t)b * S + orow) * H + h * HD, kh, q0 + r < S);
                }
            }
        }
        if (last) {
            if (!has_next) break;
            m = 0.f; l2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int d = 0; d < DB; d++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[d][e] = 0.f;
#pragma unroll
            for (int st = 0; st < KSTEPS; st++) qf[st] = qn[st];
        } else if (!has_next) {
            break;   // unreachable: a tile that is not an item's last always has a successor
        }
        it = nit; j = nj; sl ^= 1;
    }
}

// One item per workgroup, everything staged at once (the variant launched at head size 32): the same LDS-DMA tiles and
// swizzles as k_attn_s, all tiles of the item side by side in LDS, one barrier. Against k_attn: the staging is a handful of
// address computations per 1 KB piece instead of ~300 VALU instructions per wave of index arithmetic and predicated 16-byte
// copies (PMC on k_attn at 256 x 256: VALU busy 73 %, 7.3 VALU instructions per score of which only ~4 in the chunk loop).
// Round 3, MiniLM 256 x 256 (3072 items, 3 workgroups per CU), kernel-trace averages on one box with stages compiled out: whole
// kernel 64.8 us; exp replaced by an fma 60.7; no maximum logic 53.5; no P.V MFMAs 61.0; no Q.K^T MFMAs 52.2; no softmax VALU
// at all 49.5; staging + barrier only 37.9 (K alone 15.6, V^T alone 16.0); empty kernel 5.1. The 37.9 were token-major q / k
// ([T][384]: a head's key row is 64 B, half a cache line per row): with k_qkv384 writing q and k HEAD-major ([B][12][S][32],
// AttnArgs::qk_ld / qk_hs) the staging alone takes 23.0 and the QKV launch itself 66.4 instead of 73.3 us (its stores become
// 1 KiB runs) -- but this kernel only moves 64.1 -> 61.9: staging of one workgroup already runs under the others' softmax.
// PMC: VALU busy 64 %, matrix pipe 17 %, 26 VALU instructions per MFMA. A wave-uniform `first live block` flag (scalar selects
// instead of per-lane ones), the v_permlane32_swap exchange and 16-byte context stores took the block from 77 to 62 VALU
// instructions (16 of them v_exp_f32) and the kernel to 57.5 us on the box where it had been 61-62. Tried on top and not kept:
// the first block as its own instantiation inside the loop (two copies of the block: 100 B of scratch under the 80-register
// cap, 162 us); key tiles of 128 / 64 keys, each with its own counted vmcnt wait and barrier so that the first blocks run
// while the rest of the item is in flight: 60.0 / 64.3 against 57.5 -- the workgroups of a CU are already out of step, a
// barrier per tile costs more than the exposed part of one item's staging.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, (HD == 32 && NW <= 8) ? 6 : 4) void k_attn_d(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_d_body<HD, NW, LdPlain>(a, blockIdx.x, smem);
}

// =====================================================================================================================
// Round-5 experiments, compiled into libarchi_hip_dbg.so and scripts/micro/attn_bench only (AK_DBG_KERNELS). Neither beats the
// launched kernels (docs/EXPERIMENTS.md "Round 5: attention"; profiles/r05_attn_micro.txt, r05_issue_rates.txt):
//   k_attn_x  64 queries per wave on 16 x 16 x 32 tiles, a K / V^T fragment feeding four MFMAs, no running maximum
//   k_attn_p  k_attn_s' tiles with the block loop software-pipelined inside the wave by inline-asm statements
#if AK_DBG_KERNELS
// ---------------------------------------------------------------------------------------------------------------------
// k_attn_x (round 5; launched at both head sizes): 64 queries per wave on v_mfma_f32_16x16x32_bf16, 8 waves per 512 queries.
// k_attn_s (PMC, bge-base 128 x 512): matrix pipe 26 % busy, VALU 51 %, 47 % of the wave cycles parked in s_waitcnt /
// s_barrier -- every 32-key block was one serial chain (K reads -> 4 score MFMAs -> 80 VALU -> V reads -> 4 P.V MFMAs), a K / V^T
// fragment read from LDS fed ONE MFMA, and the softmax cost 7 VALU instructions per score (packed subtract, 16 maxima, packed
// adds). Here
//   * keys on M as before (a lane owns a query column), in 16 x 16 tiles: a lane holds 4 keys x 4 query groups per tile, the
//     accumulators are 4-register tuples, a fragment read from LDS feeds FOUR MFMAs (the wave's four 16-query groups);
//   * the running reference m of a query is an ESTIMATE, not the maximum: it enters the score MFMA as its C operand (four
//     registers of -m per query group, no subtraction), is set from the item's first live block and only raised when a block's
//     sum of 2^(s - m) leaves [0, 2^40] -- one compare per block instead of the maxima; fp32 sums and the bf16 P operand keep
//     their relative precision at any magnitude, so only overflow has to be fenced off. 2.6 VALU instructions per score
//     (v_exp_f32, v_add_f32, half a v_cvt_pk_bf16_f32);
//   * software pipeline by hand with ONE instance of every stage: the P.V MFMAs of block i - 1 share a basic block with the
//     exponentials of block i (16 MFMAs beside ~90 VALU), then the score MFMAs of block i + 1 run as a bare MFMA stream;
//   * blocks that hold padding (additive mask), the item's first block and a block that failed the range check go through
//     slow_block -- scores without a reference, maximum, O / l rescaled -- unpipelined (key order does not matter to the sums).
// The two score tiles of a 32-key block take the keys {0-7, 16-23} and {8-15, 24-31}: with V^T stored in vt_pos order a lane's
// eight exponentials (4 per tile) are then the k-slots its P.V B operand needs, and a lane's V^T A operand is 16 contiguous bytes.
// K tiles are XOR-swizzled (on the DMA's source address) by key & 6 (hd 64) / (key & 4) >> 1 (hd 32): every 16-lane group of a
// ds_read_b128 then covers all 64 banks for THIS access pattern (rows {0-3, 12-15} of one chunk with rows 4-11 of the next).
// Ring, tiles and the V^T swizzle are k_attn_s's.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int HD, int NW, int PIPE>
__global__ __launch_bounds__(NW * 64, (PIPE == 2 ? 1 : 2)) void k_attn_x(AttnArgs a, int nitems, int ktm) {
    constexpr int DT = HD / 16, KS = HD / 32, KROW = HD * 2, CRK = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, H = a.H, heads = a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int slot_k = ktm * KROW, slot_v = HD * ktm * 2, slot = slot_k + slot_v + ktm * 4;
    const uint32_t lds0 = lds_addr(smem);
    const int nqb = (S + NW * 64 - 1) / (NW * 64);
    const int n256 = S >> 8, rem = S & 255;
    const int ntl = n256 + __popc(rem >> 5);
    auto tile_at = [&](int j, int &k0, int &kt) __attribute__((always_inline)) {
        k0 = (j < n256 ? j : n256) << 8; kt = 256;
        if (j >= n256) {
            int jj = j - n256;
            for (int bit = 128; bit >= 32; bit >>= 1)
                if (rem & bit) {
                    if (jj == 0) { kt = bit; break; }
                    jj--; k0 += bit;
                }
        }
    };
    // K fragment of score tile T: lane (n, g) reads chunk (ks * 4 + g) of key row T * 8 + (n >> 3) * 16 + (n & 7)
    const int krow_l = ((n >> 3) << 4) | (n & 7);
    const int kx = (HD == 64 ? (g ^ (n & 6)) : (g ^ ((n & 4) >> 1))) << 4;

    auto issue = [&](int it, int j, int sl) __attribute__((always_inline)) {
        const int bh = it / nqb, b = bh / heads, h = bh - b * heads;
        int k0, kt; tile_at(j, k0, kt);
        const uint32_t sb = lds0 + sl * slot;
        const int nkp = (kt * KROW) >> 10;
        const char *kg = (const char *)(a.k + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)k0 * a.qk_ld);
        const char *vg = (const char *)(a.vt + ((int64_t)b * H + h * HD) * S + k0);
        const int lcr = 31 - __clz(kt >> 3);
        const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
        for (int p = wave; p < 2 * nkp + 1; p += NW) {
            if (p < nkp) {
                const int gi = p * 64 + lane, row = gi / CRK, c = gi % CRK;
                const int swz = HD == 64 ? (row & 6) : ((row & 4) >> 1);
                glds16(kg + (int64_t)row * a.qk_ld * 2 + ((c ^ swz) << 4), sb + p * 1024);
            } else if (p < 2 * nkp) {
                const int gi = (p - nkp) * 64 + lane, row = gi >> lcr, c = gi & ((1 << lcr) - 1);
                const int swz = (row >> vsh) & vmsk;
                glds16(vg + (int64_t)row * S * 2 + ((c ^ swz) << 4), sb + slot_k + (p - nkp) * 1024);
            } else if (lane * 4 < kt) {
                glds16(a.maskf + (int64_t)b * S + k0 + lane * 4, sb + slot_k + slot_v);
            }
        }
    };
    auto load_q = [&](int it, uint4 (&qv)[4][KS]) __attribute__((always_inline)) {
        const int bh = it / nqb, qb = it - bh * nqb, b = bh / heads, h = bh - b * heads;
#pragma unroll
        for (int qg = 0; qg < 4; qg++) {
            int qrow = qb * (NW * 64) + wave * 64 + qg * 16 + n;
            if (qrow >= S) qrow = S - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
                qv[qg][ks] = *(const uint4 *)(a.q + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)qrow * a.qk_ld + ks * 32 + g * 8);
        }
    };

    int it = blockIdx.x, j = 0, sl = 0;
    uint4 qf[4][KS];
    f32x4 o[DT][4], negm[4];
    float m[4], l[4];
    bool started = false;                              // the item has met a live block: m is set
#pragma unroll
    for (int qg = 0; qg < 4; qg++) {
        m[qg] = 0.f; l[qg] = 0.f;
        negm[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; dt++) o[dt][qg] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    issue(it, 0, 0);
    load_q(it, qf);

    while (true) {
        const uint32_t flags_raw = __builtin_amdgcn_readfirstlane(a.blkmask[(it / nqb) / heads]);
        const uint32_t flags_all = flags_raw & 0xffffu, full_all = flags_raw >> 16;
        int nit = it, nj = j + 1;
        if (nj == ntl) { nit = it + gridDim.x; nj = 0; }
        const bool has_next = nit < nitems;
        wait_vm<0>();
        __syncthreads();
        if (has_next) issue(nit, nj, sl ^ 1);
        const bool last = j == ntl - 1;

        const int bh = it / nqb, qb = it - bh * nqb, b = bh / heads, h = bh - b * heads;
        const int q0 = qb * (NW * 64) + wave * 64;
        if (q0 < S) {
            int k0, kt; tile_at(j, k0, kt);
            const char *sb = smem + sl * slot;
            const char *sV = sb + slot_k;
            const float *sM = (const float *)(sb + slot_k + slot_v);
            const int lcr = 31 - __clz(kt >> 3);
            const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
            const int kt2 = kt * 2;
            const int vx = (g ^ ((n >> vsh) & vmsk)) << 4;
            const char *krow = sb + krow_l * KROW;
            const char *vrow = sV + n * kt2;
            const int mrow = ((g >> 1) << 4) | ((g & 1) << 2);

            // score tiles of a FULL block (32 real keys), relative to the references (C = -m)
            auto qk = [&](int blk, f32x4 (&sc)[2][4]) __attribute__((always_inline)) {
                const char *kr = krow + blk * 32 * KROW;
#pragma unroll
                for (int T = 0; T < 2; T++)
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const uint4 kf = *(const uint4 *)(kr + T * 8 * KROW + ((ks << 6) ^ kx));
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) sc[T][qg] = mfma16(kf, qf[qg][ks], ks == 0 ? negm[qg] : sc[T][qg]);
                    }
            };
            auto pv = [&](int blk, const uint4 (&pb)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    const uint4 va = *(const uint4 *)(vrow + dt * 16 * kt2 + ((blk * 64) ^ vx));
#pragma unroll
                    for (int qg = 0; qg < 4; qg++) o[dt][qg] = mfma16(va, pb[qg], o[dt][qg]);
                }
            };
            // exponentials of one query group's eight scores, their sum, the packed P operand
            auto expsum = [&](const f32x4 &x0, const f32x4 &x1, uint4 &pb, float &bs) __attribute__((always_inline)) {
                float p[8];
#pragma unroll
                for (int e = 0; e < 4; e++) { p[e] = __builtin_amdgcn_exp2f(x0[e]); p[4 + e] = __builtin_amdgcn_exp2f(x1[e]); }
                bs = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                pb = uint4{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
            };
            // generic block, unpipelined: scores without a reference (+ the additive mask of a block with padding), their maximum,
            // m raised (set, on the item's first block), O and l rescaled, exponentials, P . V
            auto slow_block = [&](int blk) __attribute__((always_inline)) {
                const char *kr = krow + blk * 32 * KROW;
                const bool full = ((full_all >> (k0 >> 5)) >> blk) & 1;
                const bool first = !started;
                f32x4 acc[2][4];
#pragma unroll
                for (int T = 0; T < 2; T++) {
                    f32x4 mk = {0.f, 0.f, 0.f, 0.f};
                    if (!full) mk = *(const f32x4 *)&sM[blk * 32 + T * 8 + mrow];
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const uint4 kf = *(const uint4 *)(kr + T * 8 * KROW + ((ks << 6) ^ kx));
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) acc[T][qg] = mfma16(kf, qf[qg][ks], ks == 0 ? mk : acc[T][qg]);
                    }
                }
                uint4 pb[4];
#pragma unroll
                for (int qg = 0; qg < 4; qg++) {
                    float mx = fmaxf(fmaxf(fmaxf(acc[0][qg][0], acc[0][qg][1]), fmaxf(acc[0][qg][2], acc[0][qg][3])),
                                     fmaxf(fmaxf(acc[1][qg][0], acc[1][qg][1]), fmaxf(acc[1][qg][2], acc[1][qg][3])));
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float m_old = m[qg];
                    float m_new = first ? mx : fmaxf(m_old, mx);
                    if (!(m_new > -__builtin_inff())) m_new = 0.f;        // (cannot happen: a live block holds a real key)
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m_old - m_new);
                    l[qg] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; dt++) o[dt][qg] *= alpha;
                    float bs;
                    expsum(acc[0][qg] - m_new, acc[1][qg] - m_new, pb[qg], bs);
                    l[qg] += bs;
                    m[qg] = m_new;
                    negm[qg] = f32x4{-m_new, -m_new, -m_new, -m_new};
                }
                pv(blk, pb);
                started = true;
            };

            const uint32_t tmask = (kt >> 5) >= 32 ? ~0u : ((1u << (kt >> 5)) - 1u);
            const uint32_t lv = (flags_all >> (k0 >> 5)) & tmask;
            uint32_t lf = lv & (full_all >> (k0 >> 5));                 // blocks of 32 real keys: the pipeline
            uint32_t redo = lv & ~lf;                                   // live blocks that hold padding: slow_block
            if (!started && !redo && lf) { redo = lf & (0u - lf); lf &= lf - 1; }     // the item's first block sets m
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                while (redo) { const int blk = __builtin_ctz(redo); redo &= redo - 1; slow_block(blk); }
                if (pass || !lf) break;
                if constexpr (PIPE == 0) {
                    // unpipelined: scores, exponentials, P . V per block
                    while (lf) {
                        const int blk = __builtin_ctz(lf);
                        lf &= lf - 1;
                        f32x4 sc[2][4];
                        uint4 pb[4];
                        float bs[4];
                        qk(blk, sc);
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) expsum(sc[0][qg], sc[1][qg], pb[qg], bs[qg]);
                        const float bmax = fmaxf(fmaxf(bs[0], bs[1]), fmaxf(bs[2], bs[3]));
                        if (__any(!(bmax <= 0x1p40f))) { redo = (1u << blk) | lf; lf = 0; break; }
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) l[qg] += bs[qg];
                        pv(blk, pb);
                    }
                } else {
                    // pipelined, one instance: per query group the P.V MFMAs of block prv, the exponentials of block cur and the
                    // score MFMAs of block nxt (renewing the score registers the exponentials have just read). The first turn
                    // multiplies a zero P, the last one computes scores nobody reads.
                    f32x4 sc[2][4];
                    uint4 pb[4];
                    int cur = __builtin_ctz(lf), prv = cur;
                    lf &= lf - 1;
                    qk(cur, sc);
#pragma unroll
                    for (int qg = 0; qg < 4; qg++) pb[qg] = uint4{0, 0, 0, 0};
                    bool ok = true;
                    while (true) {
                        const int nxt = lf ? __builtin_ctz(lf) : cur;
                        uint4 va[DT], kf[2][KS];
#pragma unroll
                        for (int dt = 0; dt < DT; dt++) va[dt] = *(const uint4 *)(vrow + dt * 16 * kt2 + ((prv * 64) ^ vx));
#pragma unroll
                        for (int T = 0; T < 2; T++)
#pragma unroll
                            for (int ks = 0; ks < KS; ks++) kf[T][ks] = *(const uint4 *)(krow + (nxt * 32 + T * 8) * KROW + ((ks << 6) ^ kx));
                        float bs[4];
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) {
#pragma unroll
                            for (int dt = 0; dt < DT; dt++) o[dt][qg] = mfma16(va[dt], pb[qg], o[dt][qg]);
                            expsum(sc[0][qg], sc[1][qg], pb[qg], bs[qg]);
#pragma unroll
                            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                                for (int T = 0; T < 2; T++) sc[T][qg] = mfma16(kf[T][ks], qf[qg][ks], ks == 0 ? negm[qg] : sc[T][qg]);
                            // issue order of this query group (a wave streaming MFMAs back to back leaves the SIMD's other wave ONE
                            // VALU issue per MFMA -- scripts/micro/issue_rates -- so every MFMA is followed by this wave's own VALU
                            // work): P.V MFMAs beside the exponentials, then the score MFMAs beside the sums and conversions
#pragma unroll
                            for (int i = 0; i < DT; i++) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x402, 8 / DT, 0);
                            }
#pragma unroll
                            for (int i = 0; i < 2 * KS; i++) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x402, (11 + 2 * KS - 1) / (2 * KS), 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const float bmax = fmaxf(fmaxf(bs[0], bs[1]), fmaxf(bs[2], bs[3]));
                        if (__any(!(bmax <= 0x1p40f))) { redo = (1u << cur) | lf; lf = 0; ok = false; break; }   // nothing of cur added yet
#pragma unroll
                        for (int qg = 0; qg < 4; qg++) l[qg] += bs[qg];
                        prv = cur;
                        if (!lf) break;
                        cur = nxt;
                        lf &= lf - 1;
                    }
                    if (ok) pv(prv, pb);                                 // drain
                }
            }
            if (last) {
                // context rows: lane (query n of group qg, g) holds features 16 dt + 4 g + {0..3}; two swap stages turn the 4 x 4
                // (dt, g) pieces of a query into 32 contiguous bytes per lane (features 16 g .. 16 g + 15 at hd 64)
#pragma unroll
                for (int qg = 0; qg < 4; qg++) {
                    float lt = l[qg];
                    lt += __shfl_xor(lt, 16);
                    lt += __shfl_xor(lt, 32);
                    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
                    const int qr = q0 + qg * 16 + n;
                    const int orow = qr < S ? qr : S - 1;
                    uint16_t *dst = a.ctx + ((int64_t)b * S + orow) * H + h * HD;
                    uint32_t w[DT][2];
#pragma unroll
                    for (int dt = 0; dt < DT; dt++) {
                        w[dt][0] = pack_bf16x2(o[dt][qg][0] * inv, o[dt][qg][1] * inv);
                        w[dt][1] = pack_bf16x2(o[dt][qg][2] * inv, o[dt][qg][3] * inv);
                    }
                    if constexpr (DT == 4) {
                        // stage 1 (lanes g <-> g ^ 2): pieces dt {0,1} gather in g < 2, {2,3} in g >= 2
#pragma unroll
                        for (int dt = 0; dt < 2; dt++)
#pragma unroll
                            for (int jj = 0; jj < 2; jj++) {
                                const auto x = __builtin_amdgcn_permlane32_swap(w[dt][jj], w[dt + 2][jj], false, false);
                                w[dt][jj] = x[0]; w[dt + 2][jj] = x[1];
                            }
                        // now g < 2 holds: w[0], w[1] = own dt 0, 1 ; w[2], w[3] = dt 0, 1 of lane g + 2.  g >= 2: w[0], w[1] = dt 2, 3 of lane g - 2; w[2], w[3] own dt 2, 3
                        // stage 2 (lanes g <-> g ^ 1)
#pragma unroll
                        for (int pr = 0; pr < 4; pr += 2)
#pragma unroll
                            for (int jj = 0; jj < 2; jj++) {
                                const auto x = __builtin_amdgcn_permlane16_swap(w[pr][jj], w[pr + 1][jj], false, false);
                                w[pr][jj] = x[0]; w[pr + 1][jj] = x[1];
                            }
                        // lane g now holds the four 8-byte pieces of feature tile dt = g, from source lanes 0 .. 3 in w[0 .. 3]:
                        // features 16 g .. 16 g + 15, 32 contiguous bytes
                        if (qr < S) {
                            *(uint4 *)((char *)dst + g * 32) = uint4{w[0][0], w[0][1], w[1][0], w[1][1]};
                            *(uint4 *)((char *)dst + g * 32 + 16) = uint4{w[2][0], w[2][1], w[3][0], w[3][1]};
                        }
                    } else {
                        if (qr < S) {
#pragma unroll
                            for (int dt = 0; dt < DT; dt++) *(uint2 *)((char *)dst + dt * 32 + g * 8) = uint2{w[dt][0], w[dt][1]};
                        }
                    }
                }
            }
        }
        if (last) {
            if (!has_next) break;
            load_q(nit, qf);
            started = false;
#pragma unroll
            for (int qg = 0; qg < 4; qg++) {
                m[qg] = 0.f; l[qg] = 0.f;
                negm[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < DT; dt++) o[dt][qg] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else if (!has_next) {
            break;
        }
        it = nit; j = nj; sl ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_attn_p (round 5): k_attn_s's tiles (32 x 32 x 16 MFMA, 32 queries per wave, the same ring / swizzles), 8 waves of 32 queries
// per workgroup (256 queries: two workgroups per sequence and head at S = 512), 256 registers, and the block loop software-
// pipelined INSIDE the wave. Measured (scripts/micro/issue_rates): a wave streaming MFMAs back to back leaves the SIMD's other
// waves ONE VALU issue per MFMA, so the 4-MFMA bursts and 80-instruction softmax stretches of k_attn_s' waves serialise on a SIMD
// however many waves it holds (matrix pipe 26 % + VALU 51 % busy = the whole launch); VALU work only hides under an MFMA of the
// SAME instruction stream. One pipeline stage here = one basic block: the score MFMAs of block i + 1 and the P.V MFMAs of block
// i - 1 (8 MFMAs) with the exponentials, sums and conversions of block i between them (sched_group_barrier: 1 MFMA, 5 VALU).
// The running reference m of a query is an ESTIMATE, not the maximum (see k_attn_x): C = -m in the score MFMA, one range check
// per block, a cold generic path (slow_block) for blocks with padding, the item's first block and blocks that fail the check.
template <int HD, int NW, bool DBG = false>
__global__ __launch_bounds__(NW * 64, 2) void k_attn_p(AttnArgs a, int nitems, int ktm, long long *dbg = nullptr) {
    constexpr int DB = HD / 32, KSTEPS = HD / 16, KROW = HD * 2, CRK = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, H = a.H, heads = a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int slot_k = ktm * KROW, slot_v = HD * ktm * 2, slot = slot_k + slot_v + ktm * 4;
    const uint32_t lds0 = lds_addr(smem);
    const int nqb = (S + NW * 32 - 1) / (NW * 32);
    const int n256 = S >> 8, rem = S & 255;
    const int ntl = n256 + __popc(rem >> 5);
    auto tile_at = [&](int j, int &k0, int &kt) __attribute__((always_inline)) {
        k0 = (j < n256 ? j : n256) << 8; kt = 256;
        if (j >= n256) {
            int jj = j - n256;
            for (int bit = 128; bit >= 32; bit >>= 1)
                if (rem & bit) {
                    if (jj == 0) { kt = bit; break; }
                    jj--; k0 += bit;
                }
        }
    };
    // item index -> (sequence * heads, query block): the nqb workgroups of one (sequence, head) are 8 apart in the grid, i.e. on one
    // XCD at the same time, so the second one's K / V^T DMA hits that XCD's L2
    auto item_at = [&](int it, int &bh, int &qb) __attribute__((always_inline)) {
        if (nqb == 2) { const int x = it & 7, rr = it >> 3; qb = rr & 1; bh = ((rr >> 1) << 3) | x; }
        else { bh = it / nqb; qb = it - bh * nqb; }
    };
    const int kx = (kh ^ (HD == 64 ? (r >> 1) & 7 : (r >> 2) & 3)) << 4;

    // LDS-DMA of the next tile, one 1 KiB piece at a time: the pieces of a wave are issued between its pipeline stages (all of a
    // tile's pieces issued at once behind the barrier cost every wave ~2.7 k cycles of queueing per tile: 17 % of the launch)
    const char *dma_k = nullptr, *dma_v = nullptr, *dma_m = nullptr;
    uint32_t dma_sb = 0;
    int dma_nkp = 0, dma_kt = 0, dma_lcr = 0, dma_p = 1 << 30, dma_np = 0;
    auto issue_setup = [&](int it, int j, int sl) __attribute__((always_inline)) {
        int bh, qb; item_at(it, bh, qb);
        const int b = bh / heads, h = bh - b * heads;
        int k0, kt; tile_at(j, k0, kt);
        dma_sb = lds0 + sl * slot;
        dma_nkp = (kt * KROW) >> 10;
        dma_k = (const char *)(a.k + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)k0 * a.qk_ld);
        dma_v = (const char *)(a.vt + ((int64_t)b * H + h * HD) * S + k0);
        dma_m = (const char *)(a.maskf + (int64_t)b * S + k0);
        dma_kt = kt;
        dma_lcr = 31 - __clz(kt >> 3);
        dma_p = wave; dma_np = 2 * dma_nkp + 1;
    };
    auto glds16s = [&](const char *base, uint32_t voff, uint32_t lds) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory", "m0");
    };
    auto issue_piece = [&]() __attribute__((always_inline)) {          // the wave's next piece, if any
        const int p = dma_p;
        if (p >= dma_np) return;
        dma_p = p + NW;
        if (p < dma_nkp) {
            const int g = p * 64 + lane, row = g / CRK, c = g % CRK;
            const int swz = HD == 64 ? (row >> 1) & 7 : (row >> 2) & 3;
            glds16s(dma_k, (uint32_t)(row * (a.qk_ld * 2) + ((c ^ swz) << 4)), dma_sb + p * 1024);
        } else if (p < 2 * dma_nkp) {
            const int lcr = dma_lcr;
            const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
            const int g = (p - dma_nkp) * 64 + lane, row = g >> lcr, c = g & ((1 << lcr) - 1);
            const int swz = (row >> vsh) & vmsk;
            glds16s(dma_v, (uint32_t)(row * (S * 2) + ((c ^ swz) << 4)), dma_sb + slot_k + (p - dma_nkp) * 1024);
        } else if (lane * 4 < dma_kt) {
            glds16s(dma_m, (uint32_t)(lane * 16), dma_sb + slot_k + slot_v);
        }
    };
    auto load_q = [&](int it, uint4 (&qv)[KSTEPS]) __attribute__((always_inline)) {
        int bh, qb; item_at(it, bh, qb);
        const int b = bh / heads, h = bh - b * heads;
        int qrow = qb * (NW * 32) + wave * 32 + r;
        if (qrow >= S) qrow = S - 1;
#pragma unroll
        for (int st = 0; st < KSTEPS; st++)
            qv[st] = *(const uint4 *)(a.q + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)qrow * a.qk_ld + st * 16 + kh * 8);
    };

    long long t_wait = 0, t_issue = 0, t_slow = 0, t_pipe = 0, t_epi = 0, t_all = 0, ts = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if constexpr (DBG) return (long long)__builtin_readcyclecounter(); else return 0ll; };
    if constexpr (DBG) t_all = stamp();
    int it = blockIdx.x, j = 0, sl = 0;
    uint4 qf[KSTEPS];
    f32x16 o[DB], negm;
    float m = 0.f, l = 0.f;
    bool started = false;                              // the item has met a live block: m is set
#pragma unroll
    for (int e = 0; e < 16; e++) negm[e] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    issue_setup(it, 0, 0);
    while (dma_p < dma_np) issue_piece();
    load_q(it, qf);
    uint4 qn[KSTEPS];
#pragma unroll
    for (int st = 0; st < KSTEPS; st++) qn[st] = qf[st];

    while (true) {
        int bh, qb; item_at(it, bh, qb);
        const uint32_t flags_raw = __builtin_amdgcn_readfirstlane(a.blkmask[bh / heads]);
        const uint32_t flags_all = flags_raw & 0xffffu, full_all = flags_raw >> 16;
        int nit = it, nj = j + 1;
        if (nj == ntl) { nit = it + gridDim.x; nj = 0; }
        const bool has_next = nit < nitems;
        ts = stamp();
        wait_vm<0>();
        __syncthreads();
        if constexpr (DBG) { const long long t = stamp(); t_wait += t - ts; ts = t; }
        if (has_next) issue_setup(nit, nj, sl ^ 1);
        const bool last = j == ntl - 1;
        if (last && has_next) load_q(nit, qn);               // the next item's queries: in flight under this tile
        if constexpr (DBG) { const long long t = stamp(); t_issue += t - ts; ts = t; }

        const int b = bh / heads, h = bh - b * heads;
        const int q0 = qb * (NW * 32) + wave * 32;
        if (q0 < S) {
            int k0, kt; tile_at(j, k0, kt);
            const char *sb = smem + sl * slot;
            const char *sV = sb + slot_k;
            const float *sM = (const float *)(sb + slot_k + slot_v);
            const int lcr = 31 - __clz(kt >> 3);
            const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
            const int vx = (kh ^ ((r >> vsh) & vmsk)) << 4;
            const char *krow = sb + r * KROW;
            const char *vrow = sV + r * (kt * 2);
            const int kt2 = kt * 2;

            auto kfrag = [&](int blk, int st) __attribute__((always_inline)) { return *(const uint4 *)(krow + blk * 32 * KROW + ((st << 5) ^ kx)); };
            auto vfrag = [&](int blk, int d, int s2) __attribute__((always_inline)) { return *(const uint4 *)(vrow + d * 32 * kt2 + ((blk * 64 + s2 * 32) ^ vx)); };
            auto pv = [&](int blk, const uint4 (&pb)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int d = 0; d < DB; d++) o[d] = mfma_bf16(vfrag(blk, d, s2), pb[s2], o[d]);
            };
            auto expsum = [&](const f32x16 &x, uint4 (&pb)[2], float &bs) __attribute__((always_inline)) {
                float p[16];
#pragma unroll
                for (int e = 0; e < 16; e++) p[e] = __builtin_amdgcn_exp2f(x[e]);
                bs = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) +
                     (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
                    pb[s2] = uint4{pack_bf16x2(p[8 * s2 + 0], p[8 * s2 + 1]), pack_bf16x2(p[8 * s2 + 2], p[8 * s2 + 3]),
                                   pack_bf16x2(p[8 * s2 + 4], p[8 * s2 + 5]), pack_bf16x2(p[8 * s2 + 6], p[8 * s2 + 7])};
            };
            // generic block, unpipelined (ONE instance): scores without a reference (+ the additive mask of a block with padding),
            // their maximum, m raised (set, on the item's first block), O and l rescaled, exponentials, P . V
            auto slow_block = [&](int blk) __attribute__((always_inline)) {
                const bool full = ((full_all >> (k0 >> 5)) >> blk) & 1;
                const bool first = !started;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; e++) acc[e] = 0.f;
                if (!full) {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float4 t = *(const float4 *)&sM[blk * 32 + 8 * g + 4 * kh];
                        acc[4 * g + 0] = t.x; acc[4 * g + 1] = t.y; acc[4 * g + 2] = t.z; acc[4 * g + 3] = t.w;
                    }
                }
#pragma unroll
                for (int st = 0; st < KSTEPS; st++) acc = mfma_bf16(kfrag(blk, st), qf[st], acc);
                float mx = acc[0];
#pragma unroll
                for (int e = 1; e < 16; e++) mx = fmaxf(mx, acc[e]);
                mx = xhalf_max(mx);
                const float m_old = m;
                float m_new = first ? mx : fmaxf(m_old, mx);
                if (!(m_new > -__builtin_inff())) m_new = 0.f;        // (cannot happen: a live block holds a real key)
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m_old - m_new);
                l *= alpha;
#pragma unroll
                for (int d = 0; d < DB; d++)
#pragma unroll
                    for (int e = 0; e < 16; e++) o[d][e] *= alpha;
                f32x16 x;
#pragma unroll
                for (int e = 0; e < 16; e++) x[e] = acc[e] - m_new;
                uint4 pb[2];
                float bs;
                expsum(x, pb, bs);
                l += bs;
                m = m_new;
#pragma unroll
                for (int e = 0; e < 16; e++) negm[e] = -m_new;
                pv(blk, pb);
                started = true;
            };
            // one pipeline stage for block cur (its scores, relative to -m, in x), as four asm statements (the compiler's scheduler
            // cannot be held to an issue order; operands are scalars because inline asm cannot name a sub-register of a tuple):
            //   1, 2: the P.V MFMAs of block prv (its V^T fragments vf were read one stage ago, its P is pb) with the 16 exponentials
            //         of cur between them, four per MFMA;
            //   3, 4: the score MFMAs of block nxt (x2) with cur's conversions to bf16 and its sum between them.
            // x dies in statement 2 and x2 is born in statement 3, pb dies in 2 and the new P is born in 3 / 4: the allocator can
            // give them the same registers, the loop carries no copies. false: some 2^(s - m) of cur is out of range; nothing of cur
            // was added (pb is then garbage).
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 is a struct: not an asm register operand)
            auto stage = [&](int nxt, f32x16 &x, u32x4 (&pb)[2], const u32x4 (&vf)[2][DB], float lo_bound) __attribute__((always_inline)) {
                u32x4 kf[KSTEPS];
#pragma unroll
                for (int st = 0; st < KSTEPS; st++) kf[st] = __builtin_bit_cast(u32x4, kfrag(nxt, st));
                float e0, e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11, e12, e13, e14, e15;
                if constexpr (DB == 2) {
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[o0], %[va], %[p], %[o0]\n\t"
                        "v_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]\n\tv_exp_f32 %[e2], %[x2]\n\tv_exp_f32 %[e3], %[x3]\n\t"
                        "v_mfma_f32_32x32x16_bf16 %[o1], %[vb], %[p], %[o1]\n\t"
                        "v_exp_f32 %[e4], %[x4]\n\tv_exp_f32 %[e5], %[x5]\n\tv_exp_f32 %[e6], %[x6]\n\tv_exp_f32 %[e7], %[x7]"
                        : [o0] "+v"(o[0]), [o1] "+v"(o[DB - 1]), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3),
                          [e4] "=&v"(e4), [e5] "=&v"(e5), [e6] "=&v"(e6), [e7] "=&v"(e7)
                        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]),
                          [x7] "v"(x[7]), [va] "v"(vf[0][0]), [vb] "v"(vf[0][DB - 1]), [p] "v"(pb[0]));
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[o0], %[va], %[p], %[o0]\n\t"
                        "v_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]\n\tv_exp_f32 %[e2], %[x2]\n\tv_exp_f32 %[e3], %[x3]\n\t"
                        "v_mfma_f32_32x32x16_bf16 %[o1], %[vb], %[p], %[o1]\n\t"
                        "v_exp_f32 %[e4], %[x4]\n\tv_exp_f32 %[e5], %[x5]\n\tv_exp_f32 %[e6], %[x6]\n\tv_exp_f32 %[e7], %[x7]"
                        : [o0] "+v"(o[0]), [o1] "+v"(o[DB - 1]), [e0] "=&v"(e8), [e1] "=&v"(e9), [e2] "=&v"(e10), [e3] "=&v"(e11),
                          [e4] "=&v"(e12), [e5] "=&v"(e13), [e6] "=&v"(e14), [e7] "=&v"(e15)
                        : [x0] "v"(x[8]), [x1] "v"(x[9]), [x2] "v"(x[10]), [x3] "v"(x[11]), [x4] "v"(x[12]), [x5] "v"(x[13]), [x6] "v"(x[14]),
                          [x7] "v"(x[15]), [va] "v"(vf[1][0]), [vb] "v"(vf[1][DB - 1]), [p] "v"(pb[1]));
                } else {
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[o0], %[va], %[p], %[o0]\n\t"
                        "v_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]\n\tv_exp_f32 %[e2], %[x2]\n\tv_exp_f32 %[e3], %[x3]\n\t"
                        "v_exp_f32 %[e4], %[x4]\n\tv_exp_f32 %[e5], %[x5]\n\tv_exp_f32 %[e6], %[x6]\n\tv_exp_f32 %[e7], %[x7]"
                        : [o0] "+v"(o[0]), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3),
                          [e4] "=&v"(e4), [e5] "=&v"(e5), [e6] "=&v"(e6), [e7] "=&v"(e7)
                        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]),
                          [x7] "v"(x[7]), [va] "v"(vf[0][0]), [p] "v"(pb[0]));
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[o0], %[va], %[p], %[o0]\n\t"
                        "v_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]\n\tv_exp_f32 %[e2], %[x2]\n\tv_exp_f32 %[e3], %[x3]\n\t"
                        "v_exp_f32 %[e4], %[x4]\n\tv_exp_f32 %[e5], %[x5]\n\tv_exp_f32 %[e6], %[x6]\n\tv_exp_f32 %[e7], %[x7]"
                        : [o0] "+v"(o[0]), [e0] "=&v"(e8), [e1] "=&v"(e9), [e2] "=&v"(e10), [e3] "=&v"(e11),
                          [e4] "=&v"(e12), [e5] "=&v"(e13), [e6] "=&v"(e14), [e7] "=&v"(e15)
                        : [x0] "v"(x[8]), [x1] "v"(x[9]), [x2] "v"(x[10]), [x3] "v"(x[11]), [x4] "v"(x[12]), [x5] "v"(x[13]), [x6] "v"(x[14]),
                          [x7] "v"(x[15]), [va] "v"(vf[1][0]), [p] "v"(pb[1]));
                }
                f32x16 x2;
                uint32_t c0, c1, c2, c3, c4, c5, c6, c7;
                const u32x4 q0 = __builtin_bit_cast(u32x4, qf[0]), q1 = __builtin_bit_cast(u32x4, qf[1]);
                if constexpr (KSTEPS == 4) {
                    const u32x4 q2 = __builtin_bit_cast(u32x4, qf[KSTEPS - 2]), q3 = __builtin_bit_cast(u32x4, qf[KSTEPS - 1]);
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[ka], %[qa], %[nm]\n\t"
                        "v_cvt_pk_bf16_f32 %[c0], %[e0], %[e1]\n\tv_cvt_pk_bf16_f32 %[c1], %[e2], %[e3]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e1]\n\tv_add_f32 %[e2], %[e2], %[e3]\n\t"
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[kb], %[qb], %[xn]\n\t"
                        "v_cvt_pk_bf16_f32 %[c2], %[e4], %[e5]\n\tv_cvt_pk_bf16_f32 %[c3], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e4], %[e4], %[e5]\n\tv_add_f32 %[e6], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e2]\n\tv_add_f32 %[e4], %[e4], %[e6]\n\tv_add_f32 %[e0], %[e0], %[e4]"
                        : [xn] "=&v"(x2), [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2),
                          [e3] "+v"(e3), [e4] "+v"(e4), [e5] "+v"(e5), [e6] "+v"(e6), [e7] "+v"(e7)
                        : [ka] "v"(kf[0]), [kb] "v"(kf[1]), [qa] "v"(q0), [qb] "v"(q1), [nm] "v"(negm));
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[ka], %[qa], %[xn]\n\t"
                        "v_cvt_pk_bf16_f32 %[c0], %[e0], %[e1]\n\tv_cvt_pk_bf16_f32 %[c1], %[e2], %[e3]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e1]\n\tv_add_f32 %[e2], %[e2], %[e3]\n\t"
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[kb], %[qb], %[xn]\n\t"
                        "v_cvt_pk_bf16_f32 %[c2], %[e4], %[e5]\n\tv_cvt_pk_bf16_f32 %[c3], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e4], %[e4], %[e5]\n\tv_add_f32 %[e6], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e2]\n\tv_add_f32 %[e4], %[e4], %[e6]\n\tv_add_f32 %[e0], %[e0], %[e4]"
                        : [xn] "+v"(x2), [c0] "=&v"(c4), [c1] "=&v"(c5), [c2] "=&v"(c6), [c3] "=&v"(c7), [e0] "+v"(e8), [e1] "+v"(e9), [e2] "+v"(e10),
                          [e3] "+v"(e11), [e4] "+v"(e12), [e5] "+v"(e13), [e6] "+v"(e14), [e7] "+v"(e15)
                        : [ka] "v"(kf[KSTEPS - 2]), [kb] "v"(kf[KSTEPS - 1]), [qa] "v"(q2), [qb] "v"(q3));
                } else {
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[ka], %[qa], %[nm]\n\t"
                        "v_cvt_pk_bf16_f32 %[c0], %[e0], %[e1]\n\tv_cvt_pk_bf16_f32 %[c1], %[e2], %[e3]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e1]\n\tv_add_f32 %[e2], %[e2], %[e3]\n\t"
                        "v_cvt_pk_bf16_f32 %[c2], %[e4], %[e5]\n\tv_cvt_pk_bf16_f32 %[c3], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e4], %[e4], %[e5]\n\tv_add_f32 %[e6], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e2]\n\tv_add_f32 %[e4], %[e4], %[e6]\n\tv_add_f32 %[e0], %[e0], %[e4]"
                        : [xn] "=&v"(x2), [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2),
                          [e3] "+v"(e3), [e4] "+v"(e4), [e5] "+v"(e5), [e6] "+v"(e6), [e7] "+v"(e7)
                        : [ka] "v"(kf[0]), [qa] "v"(q0), [nm] "v"(negm));
                    asm volatile(
                        "v_mfma_f32_32x32x16_bf16 %[xn], %[ka], %[qa], %[xn]\n\t"
                        "v_cvt_pk_bf16_f32 %[c0], %[e0], %[e1]\n\tv_cvt_pk_bf16_f32 %[c1], %[e2], %[e3]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e1]\n\tv_add_f32 %[e2], %[e2], %[e3]\n\t"
                        "v_cvt_pk_bf16_f32 %[c2], %[e4], %[e5]\n\tv_cvt_pk_bf16_f32 %[c3], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e4], %[e4], %[e5]\n\tv_add_f32 %[e6], %[e6], %[e7]\n\t"
                        "v_add_f32 %[e0], %[e0], %[e2]\n\tv_add_f32 %[e4], %[e4], %[e6]\n\tv_add_f32 %[e0], %[e0], %[e4]"
                        : [xn] "+v"(x2), [c0] "=&v"(c4), [c1] "=&v"(c5), [c2] "=&v"(c6), [c3] "=&v"(c7), [e0] "+v"(e8), [e1] "+v"(e9), [e2] "+v"(e10),
                          [e3] "+v"(e11), [e4] "+v"(e12), [e5] "+v"(e13), [e6] "+v"(e14), [e7] "+v"(e15)
                        : [ka] "v"(kf[KSTEPS - 1]), [qa] "v"(q1));
                }
                const float bs = e0 + e8;
                pb[0] = u32x4{c0, c1, c2, c3};
                pb[1] = u32x4{c4, c5, c6, c7};
                x = x2;
                const bool ok = !__any(!(bs <= 0x1p40f && bs >= lo_bound));
                if (ok) l += bs;
                return ok;
            };
            auto vfrags = [&](int blk, u32x4 (&vf)[2][DB]) __attribute__((always_inline)) {
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int d = 0; d < DB; d++) vf[s2][d] = __builtin_bit_cast(u32x4, vfrag(blk, d, s2));
            };

            const uint32_t tmask = (kt >> 5) >= 32 ? ~0u : ((1u << (kt >> 5)) - 1u);
            const uint32_t lv = (flags_all >> (k0 >> 5)) & tmask;
            uint32_t lf = lv & (full_all >> (k0 >> 5));                 // blocks of 32 real keys: the pipeline
            uint32_t redo = lv & ~lf;                                   // live blocks that hold padding: slow_block
            // (the item's first block, when it is full, runs in the pipeline against m = 0 like any other: its sum only has to lie in
            // [2^-40, 2^40] -- both sides checked -- else it is handed to slow_block, which then sets m)
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                while (redo) { const int blk = __builtin_ctz(redo); redo &= redo - 1; slow_block(blk); }
                if constexpr (DBG) { const long long t = stamp(); t_slow += t - ts; ts = t; }
                if (pass || !lf) break;
                f32x16 x;
                u32x4 pb[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, vf[2][DB];
                int cur = __builtin_ctz(lf);
                lf &= lf - 1;
                x = mfma_bf16(kfrag(cur, 0), qf[0], negm);
#pragma unroll
                for (int st = 1; st < KSTEPS; st++) x = mfma_bf16(kfrag(cur, st), qf[st], x);
                vfrags(cur, vf);                                              // first stage: a zero P against cur's own V^T
                bool ok = true;
                // stages: the P.V of block i - 1 runs in stage i; the last stage computes its own block's scores again (nobody reads them)
                while (true) {
                    const int nxt = lf ? __builtin_ctz(lf) : cur;
                    u32x4 vn[2][DB];
                    vfrags(cur, vn);                                          // for the next stage (or the drain)
                    issue_piece();
                    ok = stage(nxt, x, pb, vf, started ? 0.f : 0x1p-40f);
                    started = started || ok;
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                        for (int d = 0; d < DB; d++) vf[s2][d] = vn[s2][d];
                    if (!ok || !lf) break;
                    cur = nxt; lf &= lf - 1;
                }
                if (!ok) { redo = (1u << cur) | lf; lf = 0; continue; }       // (the previous block's P . V has been issued; cur and the rest: slow_block)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)                                // drain: P . V of the last block
#pragma unroll
                    for (int d = 0; d < DB; d++)
                        o[d] = mfma_bf16(__builtin_bit_cast(uint4, vf[s2][d]), __builtin_bit_cast(uint4, pb[s2]), o[d]);
                if constexpr (DBG) { const long long t = stamp(); t_pipe += t - ts; ts = t; }
            }
            if (last) {
                float lt = l;
                lt += __shfl_xor(lt, 32);
                const float inv = lt > 0.f ? 1.0f / lt : 0.f;
                const int orow = q0 + r < S ? q0 + r : S - 1;
                store_ctx_rows<DB>(o, inv, a.ctx + ((int64_t)b * S + orow) * H + h * HD, kh, q0 + r < S);
            }
        }
        while (dma_p < dma_np) issue_piece();                 // what the stages did not take
        if (last) {
            if (!has_next) break;
#pragma unroll
            for (int st = 0; st < KSTEPS; st++) qf[st] = qn[st];
            if constexpr (DBG) { const long long t = stamp(); t_epi += t - ts; ts = t; }
            started = false;
            m = 0.f; l = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) negm[e] = 0.f;
#pragma unroll
            for (int d = 0; d < DB; d++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[d][e] = 0.f;
        } else if (!has_next) {
            break;
        }
        it = nit; j = nj; sl ^= 1;
    }
    if constexpr (DBG) {
        if (lane == 0 && dbg) {
            long long *d = dbg + ((int64_t)blockIdx.x * NW + wave) * 8;
            d[0] = stamp() - t_all; d[1] = t_wait; d[2] = t_issue; d[3] = t_slow; d[4] = t_pipe; d[5] = t_epi;
        }
    }
}

#endif  // AK_DBG_KERNELS

int launch_attn(const AttnArgs &a0, hipStream_t st) {
    AttnArgs a = a0;
    const int hd = a.H / a.heads;
    if (a.qk_ld == 0) { a.qk_ld = a.H; a.qk_hs = hd; }        // token-major q / k
    if (hd != 32 && hd != 64) AK_FAIL(-1, "attention: head size must be 32 or 64");
    if (a.S % 32 || a.S > 512) AK_FAIL(-1, "attention: S must be a multiple of 32 and <= 512");
    size_t lds = (size_t)a.S * (hd * 2 + 16) + (size_t)hd * (a.S * 2 + 16) + (size_t)a.S * 4;
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    static const int force_nw = env_get("AK_ATTN_NW") ? atoi(env_get("AK_ATTN_NW")) : 0;
    // AK_ATTN_STREAM=0 / 1 / 2 forces k_attn / k_attn_s / k_attn_d (A/B). Default: k_attn_s at hd = 64, k_attn_d at hd = 32. On full
    // masks the three are within 1 % of each other (MiniLM 256 x 256: 2.13-2.16 ms per forward; bge-base 128 x 512: 15.5-15.6 vs
    // 15.6-15.8 for k_attn); on padded batches (real lengths uniform in [32, S]) the two DMA-staged kernels skip the 32-key
    // blocks that hold only padding: MiniLM 2.09 vs 2.13 ms, bge-base 15.45 vs 16.05 ms.
    static const int force_stream = env_get("AK_ATTN_STREAM") ? atoi(env_get("AK_ATTN_STREAM")) : (env_get("AK_ATTN_OLD") ? 0 : -1);
    const int variant = force_stream >= 0 ? force_stream : (hd == 64 ? 1 : 2);
    if (variant == 2 && a.maskf && a.blkmask) {
        static std::atomic<bool> attr_d{false};
        if (!attr_d) {
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_d<64, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_d = true;
        }
        const int nw = force_nw ? force_nw : (a.S >= 512 ? 16 : (a.S >= 256 ? 8 : 4));
        const size_t ldsd = (size_t)a.S * (hd * 2 + 2 * hd + 4);
        const int nitems = a.B * a.heads * ((a.S + nw * 32 - 1) / (nw * 32));
        if (hd == 32) {
            if (nw == 16) k_attn_d<32, 16><<<nitems, 1024, ldsd, st>>>(a);
            else if (nw == 8) k_attn_d<32, 8><<<nitems, 512, ldsd, st>>>(a);
            else k_attn_d<32, 4><<<nitems, 256, ldsd, st>>>(a);
        } else {
            if (nw == 16) k_attn_d<64, 16><<<nitems, 1024, ldsd, st>>>(a);
            else if (nw == 8) k_attn_d<64, 8><<<nitems, 512, ldsd, st>>>(a);
            else k_attn_d<64, 4><<<nitems, 256, ldsd, st>>>(a);
        }
        AK_HIP(hipGetLastError());
        return 0;
    }
#if AK_DBG_KERNELS
    if (variant == 3 && a.maskf && a.blkmask) {
        static const int pipe = env_get("AK_ATTN_PIPE") ? atoi(env_get("AK_ATTN_PIPE")) : 1;
        static std::atomic<int> cus_x{0};
        if (!cus_x) {
            int dev = 0; hipDeviceProp_t pr;
            AK_HIP(hipGetDevice(&dev)); AK_HIP(hipGetDeviceProperties(&pr, dev));
#define AK_ATTN_ATTR(HDV, NWV, PV) AK_HIP(hipFuncSetAttribute((const void *)k_attn_x<HDV, NWV, PV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define AK_ATTN_ATTRS(HDV, NWV) AK_ATTN_ATTR(HDV, NWV, 0); AK_ATTN_ATTR(HDV, NWV, 1); AK_ATTN_ATTR(HDV, NWV, 2)
            AK_ATTN_ATTRS(32, 1); AK_ATTN_ATTRS(32, 2); AK_ATTN_ATTRS(32, 4); AK_ATTN_ATTR(32, 8, 0); AK_ATTN_ATTR(32, 8, 1);
            AK_ATTN_ATTRS(64, 1); AK_ATTN_ATTRS(64, 2); AK_ATTN_ATTRS(64, 4); AK_ATTN_ATTR(64, 8, 0); AK_ATTN_ATTR(64, 8, 1);
#undef AK_ATTN_ATTRS
#undef AK_ATTN_ATTR
            cus_x = pr.multiProcessorCount;
        }
        int nw = force_nw ? force_nw : (a.S > 256 ? 8 : (a.S > 128 ? 4 : (a.S > 64 ? 2 : 1)));
        if (pipe == 2 && nw > 4) nw = 4;                      // one wave per SIMD
        int ktm = a.S >= 256 ? 256 : 32;
        if (a.S < 256) while (ktm * 2 <= a.S) ktm *= 2;
        const size_t slot = (size_t)ktm * hd * 2 * 2 + (size_t)ktm * 4;
        const size_t ring = 2 * slot;
        const int nqb = (a.S + nw * 64 - 1) / (nw * 64);
        const int nitems = a.B * a.heads * nqb;
        int per_cu = (pipe == 2 ? 4 : 8) / nw;                // 8 waves per CU: two per SIMD at 256 registers
        while (per_cu > 1 && per_cu * ring > 160 * 1024) per_cu--;
        const int grid = nitems < cus_x * per_cu ? nitems : cus_x * per_cu;
#define AK_ATTN_X(HDV, NWV, PV) k_attn_x<HDV, NWV, PV><<<grid, NWV * 64, ring, st>>>(a, nitems, ktm)
#define AK_ATTN_XP(HDV, NWV) do { if (pipe == 2) AK_ATTN_X(HDV, NWV, 2); else if (pipe) AK_ATTN_X(HDV, NWV, 1); else AK_ATTN_X(HDV, NWV, 0); } while (0)
#define AK_ATTN_XP8(HDV) do { if (pipe) AK_ATTN_X(HDV, 8, 1); else AK_ATTN_X(HDV, 8, 0); } while (0)
        if (hd == 32) {
            if (nw == 8) AK_ATTN_XP8(32); else if (nw == 4) AK_ATTN_XP(32, 4); else if (nw == 2) AK_ATTN_XP(32, 2); else AK_ATTN_XP(32, 1);
        } else {
            if (nw == 8) AK_ATTN_XP8(64); else if (nw == 4) AK_ATTN_XP(64, 4); else if (nw == 2) AK_ATTN_XP(64, 2); else AK_ATTN_XP(64, 1);
        }
#undef AK_ATTN_XP8
#undef AK_ATTN_XP
#undef AK_ATTN_X
        AK_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 4 && a.maskf && a.blkmask) {
        static std::atomic<int> cus_p{0};
        if (!cus_p) {
            int dev = 0; hipDeviceProp_t pr;
            AK_HIP(hipGetDevice(&dev)); AK_HIP(hipGetDeviceProperties(&pr, dev));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_p<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            cus_p = pr.multiProcessorCount;
        }
        const int nw = force_nw ? force_nw : (a.S > 128 ? 8 : (a.S > 64 ? 4 : (a.S > 32 ? 2 : 1)));
        int ktm = a.S >= 256 ? 256 : 32;
        if (a.S < 256) while (ktm * 2 <= a.S) ktm *= 2;
        const size_t slot = (size_t)ktm * hd * 2 * 2 + (size_t)ktm * 4;
        const size_t ring = 2 * slot;
        const int nqb = (a.S + nw * 32 - 1) / (nw * 32);
        const int nitems = a.B * a.heads * nqb;
        int per_cu = 8 / nw;                                  // 8 waves per CU: two per SIMD at 256 registers
        while (per_cu > 1 && per_cu * ring > 160 * 1024) per_cu--;
        int grid = nitems < cus_p * per_cu ? nitems : cus_p * per_cu;
        if (nqb == 2) {                                       // the item map pairs workgroups 8 apart: whole groups of 16 items
            if (((a.B * a.heads) & 7) || (grid & 15)) AK_FAIL(-1, "attention (k_attn_p): sequences * heads must be a multiple of 8");
        }
        static long long *dbgbuf = nullptr;
        static const bool dbg_on = env_get("AK_ATTN_DBG") != nullptr;
        if (dbg_on && !dbgbuf) { AK_HIP(hipMalloc((void **)&dbgbuf, (size_t)2048 * 8 * 8 * 8)); AK_HIP(hipMemset(dbgbuf, 0, (size_t)2048 * 8 * 8 * 8)); }
#define AK_ATTN_P(HDV, NWV) do { if (dbg_on) k_attn_p<HDV, NWV, true><<<grid, NWV * 64, ring, st>>>(a, nitems, ktm, dbgbuf); else k_attn_p<HDV, NWV><<<grid, NWV * 64, ring, st>>>(a, nitems, ktm); } while (0)
        if (hd == 32) {
            if (nw == 8) AK_ATTN_P(32, 8); else if (nw == 4) AK_ATTN_P(32, 4); else if (nw == 2) AK_ATTN_P(32, 2); else AK_ATTN_P(32, 1);
        } else {
            if (nw == 8) AK_ATTN_P(64, 8); else if (nw == 4) AK_ATTN_P(64, 4); else if (nw == 2) AK_ATTN_P(64, 2); else AK_ATTN_P(64, 1);
        }
#undef AK_ATTN_P
        AK_HIP(hipGetLastError());
        if (dbg_on) {
            AK_HIP(hipStreamSynchronize(st));
            std::vector<long long> hd_((size_t)grid * nw * 8);
            AK_HIP(hipMemcpy(hd_.data(), dbgbuf, hd_.size() * 8, hipMemcpyDeviceToHost));
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < (size_t)grid * nw; i++) for (int c = 0; c < 6; c++) acc[c] += (double)hd_[i * 8 + c];
            fprintf(stderr, "k_attn_p cycles per wave: all %.0f wait+barrier %.0f issue %.0f slow %.0f pipe %.0f epilogue+q %.0f (grid %d x %d waves, %d items)\n",
                    acc[0] / (grid * nw), acc[1] / (grid * nw), acc[2] / (grid * nw), acc[3] / (grid * nw), acc[4] / (grid * nw), acc[5] / (grid * nw), grid, nw, nitems);
        }
        return 0;
    }
#endif
    const bool stream = variant == 1;
    if (stream && a.maskf && a.blkmask) {
        static int cus = 0;
        if (!cus) {
            int dev = 0; hipDeviceProp_t pr;
            AK_HIP(hipGetDevice(&dev)); AK_HIP(hipGetDeviceProperties(&pr, dev));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AK_HIP(hipFuncSetAttribute((const void *)k_attn_s<64, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            cus = pr.multiProcessorCount;      // last: a second thread that sees it set may launch at once
        }
        const int nw = force_nw ? force_nw : (a.S >= 512 ? 16 : (a.S >= 256 ? 8 : 4));
        int ktm = a.S >= 256 ? 256 : 32;
        if (a.S < 256) while (ktm * 2 <= a.S) ktm *= 2;       // largest power-of-two tile of this S
        const size_t slot = (size_t)ktm * hd * 2 * 2 + (size_t)ktm * 4;
        const size_t ring = 2 * slot;
        const int nqb = (a.S + nw * 32 - 1) / (nw * 32);
        const int nitems = a.B * a.heads * nqb;
        int per_cu = 16 / nw;                                 // 16 waves per CU: four per SIMD at <= 128 registers
        while (per_cu > 1 && per_cu * ring > 160 * 1024) per_cu--;
        const int grid = nitems < cus * per_cu ? nitems : cus * per_cu;
        if (hd == 32) {
            if (nw == 16) k_attn_s<32, 16><<<grid, 1024, ring, st>>>(a, nitems, ktm);
            else if (nw == 8) k_attn_s<32, 8><<<grid, 512, ring, st>>>(a, nitems, ktm);
            else k_attn_s<32, 4><<<grid, 256, ring, st>>>(a, nitems, ktm);
        } else {
            if (nw == 16) k_attn_s<64, 16><<<grid, 1024, ring, st>>>(a, nitems, ktm);
            else if (nw == 8) k_attn_s<64, 8><<<grid, 512, ring, st>>>(a, nitems, ktm);
            else k_attn_s<64, 4><<<grid, 256, ring, st>>>(a, nitems, ktm);
        }
        AK_HIP(hipGetLastError());
        return 0;
    }
    const int nw = force_nw ? force_nw : (a.S >= 512 ? 16 : (a.S >= 256 ? 8 : 4));
    dim3 grid((a.S + nw * 32 - 1) / (nw * 32), a.heads, a.B);
    if (hd == 32) {
        if (nw == 16) k_attn<32, 16><<<grid, 1024, lds, st>>>(a);
        else if (nw == 8) k_attn<32, 8, 2><<<grid, 512, lds, st>>>(a);   // 64-key chunks: 78 registers, 3 workgroups per CU (72 vs 76 us)
        else k_attn<32, 4><<<grid, 256, lds, st>>>(a);
    } else {
        if (nw == 16) k_attn<64, 16, 2><<<grid, 1024, lds, st>>>(a);   // 64-key chunks: no spill at 128 registers (208 vs 216 us)
        else if (nw == 8) k_attn<64, 8><<<grid, 512, lds, st>>>(a);
        else k_attn<64, 4><<<grid, 256, lds, st>>>(a);
    }
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_attn_prepare(const int *mask, int B, int S, float *maskf, uint32_t *blkmask, hipStream_t st) {
    k_attn_prepare<<<B, 512, 0, st>>>(mask, S, maskf, blkmask);
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
