// common.h -- shared host/device helpers for libarchi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/archi_knn.h"

namespace ak {

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string &msg);
// hipSetDevice is per host thread: every entry point that touches the GPU binds the calling thread to the process's device
// (ak_init's) first -- request / helper threads of a rank > 0 process would otherwise run on device 0. 0 on success.
int bind_thread();
#define AK_BIND()                          \
    do {                                   \
        if (ak::bind_thread()) return -10; \
    } while (0)
#define AK_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            ak::set_error(std::string(#call) + " failed: " + hipGetErrorString(e__) + " (" + \
                          __FILE__ + ":" + std::to_string(__LINE__) + ")");                  \
            return -10;                                                                      \
        }                                                                                    \
    } while (0)
#define AK_FAIL(code, msg)  \
    do {                    \
        ak::set_error(msg); \
        return (code);      \
    } while (0)

// roctx range (rocprofv3 --marker-trace shows it around the kernels of one call): the marker library is looked up at run
// time (librocprofiler-sdk-roctx / libroctx64) and the range is a no-op when neither is there. SURVEY.md section 5.
struct RoctxRange {
    explicit RoctxRange(const char *name);
    ~RoctxRange();
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
    bool on;
};

// Measurement instantiations (per-phase cycle stamps, compile-time ablations: AK_SCAN_DBG / AK_SCAN_ABLATE / AK_FFN_DBG /
// AK_FFN_ABLATE) are compiled only into libarchi_hip_dbg.so (`make dbg`: the same sources with -DAK_DBG_KERNELS=1); the
// product library carries none of them and refuses those switches. archi_amd/_lib.py loads the dbg library when one of the
// switches is set and the file exists.
#ifndef AK_DBG_KERNELS
#define AK_DBG_KERNELS 0
#endif
constexpr bool DBG_KERNELS = AK_DBG_KERNELS != 0;

constexpr int WAVE = 64;
constexpr uint64_t KEY_INVALID = ~0ull;

// ---- storage dtype <-> fp32 (bit exact, RNE) -------------------------------
__host__ __device__ inline uint32_t f32_bits(float f) {
    union { float f; uint32_t u; } v; v.f = f; return v.u;
}
__host__ __device__ inline float bits_f32(uint32_t u) {
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ inline float f16_to_f32(uint16_t h) {
    _Float16 x; __builtin_memcpy(&x, &h, 2); return (float)x;
}
__device__ inline uint16_t f32_to_f16(float f) {
    _Float16 x = (_Float16)f;  // v_cvt_f16_f32, RNE
    uint16_t h; __builtin_memcpy(&h, &x, 2); return h;
}

template <int DT> struct Store;  // DT = AK_DTYPE_*
template <> struct Store<AK_DTYPE_F32> {
    using T = float;
    static __device__ inline float load(const T *p, int64_t i) { return p[i]; }
    static __device__ inline T cvt(float f) { return f; }
};
template <> struct Store<AK_DTYPE_BF16> {
    using T = uint16_t;
    static __device__ inline float load(const T *p, int64_t i) { return bf16_to_f32(p[i]); }
    static __device__ inline T cvt(float f) { return f32_to_bf16(f); }
};
template <> struct Store<AK_DTYPE_F16> {
    using T = uint16_t;
    static __device__ inline float load(const T *p, int64_t i) { return f16_to_f32(p[i]); }
    static __device__ inline T cvt(float f) { return f32_to_f16(f); }
};
inline int dtype_size(int dt) { return dt == AK_DTYPE_F32 ? 4 : 2; }

// ---- order-preserving integer keys ----------------------------------------
// double distance -> u64, ascending key == ascending distance, every NaN maps
// to one key above +inf (Postgres float8 ordering), -0.0 == +0.0.
__host__ __device__ inline uint64_t dist_key(double d) {
    if (d != d) return 0xfff8000000000000ull;  // canonical NaN, sorts last (below KEY_INVALID)
    if (d == 0.0) d = 0.0;
    union { double d; uint64_t u; } v; v.d = d;
    return (v.u >> 63) ? ~v.u : (v.u | 0x8000000000000000ull);
}
__host__ __device__ inline double key_dist(uint64_t k) {
    union { double d; uint64_t u; } v;
    if (k == 0xfff8000000000000ull) { v.u = 0x7ff8000000000000ull; return v.d; }
    v.u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return v.d;
}
// float score (higher = better) -> u32, ascending key == descending score.
__host__ __device__ inline uint32_t score_key(float s) {
    uint32_t u = f32_bits(s);
    u = (u >> 31) ? ~u : (u | 0x80000000u);  // ascending with s
    return ~u;
}
__host__ __device__ inline float key_score(uint32_t k) {
    uint32_t u = ~k;
    u = (u >> 31) ? (u & 0x7fffffffu) : ~u;
    return bits_f32(u);
}

// ---- Philox4x32-10 + the synthetic element map (oracle/knn_oracle.c) -------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__host__ __device__ inline int bytesum(uint32_t w) {
    return (int)(w & 0xff) + (int)((w >> 8) & 0xff) + (int)((w >> 16) & 0xff) + (int)(w >> 24);
}

}  // namespace ak
