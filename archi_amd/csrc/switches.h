// switches.h -- the measurement / A-B switches of libarchi_hip.so in ONE table, read from a snapshot of the environment taken ONCE (at dlopen) instead of getenv() calls on call paths (glibc getenv racing a setenv from another thread is
// undefined behaviour; round-4 review). Tests and probe scripts that want to change a switch inside a running process call
// ak_debug_set(name, value) (archi_amd._lib.debug_set), which writes the same table.
// Switches that produce WRONG RESULTS (stage-skipping ablations: AK_SCAN_ABLATE, AK_TAIL_ABLATE, AK_QKV_DBG, AK_GEMM_ABLATE,
// AK_FFN_ABLATE, AK_ENC_NOFFN) exist only in libarchi_hip_dbg.so (AK_DBG_KERNELS): the product library neither reads nor accepts them.
#pragma once
#include <atomic>

#include "common.h"

namespace ak {

struct Switches {
    // scan plan (scan.hip)
    std::atomic<int> scan_cfg{0};        // AK_SCAN_CFG: forced tile, its letter (0 = the plan's choice)
    std::atomic<int> scan_blocks{0};     // AK_SCAN_BLOCKS: workgroups per launch (0 = resident count)
    std::atomic<int> scan_r192_pm{850};  // AK_SCAN_R192: relative cost of a 192-query pass, per mille
    std::atomic<int> scan_no192{0};      // AK_SCAN_NO192
    std::atomic<int> seed_ratio{32};     // AK_SEED_RATIO
    std::atomic<int> seed_div{0};        // AK_SEED_DIV (0 = by shard size)
    std::atomic<int> pre_div{0};         // AK_PRE_DIV (0 = by plan)
    std::atomic<int> scan_noseed{0};     // AK_SCAN_NOSEED
    std::atomic<int> scan_nopre{0};      // AK_SCAN_NOPRE
    std::atomic<int> tail_old{0};        // AK_TAIL_OLD: the three-kernel tail
    std::atomic<int> scan_dbg{0};        // AK_SCAN_DBG: cycle stamps (instrumented kernels: dbg library)
    std::atomic<int> coalesce_stats{0};  // AK_COALESCE_STATS
    std::atomic<int> query_fused{0};     // AK_QUERY_FUSED: 1 = embed_query through query_forward.hip's single launch (measured SLOWER than the 47
                                         // launches: opt-in), 2 = required (tests: fail instead of falling back), 0 = off (default)
    std::atomic<int> shard_inject{0};    // AK_SHARD_INJECT: error injection of ak_index_search_sharded_dev (shardcomm.hip; errors only)
    // WRONG RESULTS, dbg library only (always 0 in the product library)
    std::atomic<int> scan_ablate{0};     // AK_SCAN_ABLATE
    std::atomic<int> tail_ablate{0};     // AK_TAIL_ABLATE
};
Switches &switches();
// 0 on success, -1 for a name this library does not know (or a WRONG-RESULTS switch in the product library)
int switches_set(const char *name, const char *value);
// The AK_* part of the environment is SNAPSHOT once (ak_init, i.e. at library load; first use otherwise) and every later read --
// the table above and the function-local statics of the encoder's kernel selection -- comes from the snapshot: no getenv() on a
// call path, whatever thread a first forward or search happens on. env_get: the value in the snapshot or NULL.
const char *env_get(const char *name);
// integer value of a snapshot variable; wrong-result switches go through dbg_env_int, which is the constant `dflt` in the product library
int env_int(const char *name, int dflt);
inline int dbg_env_int(const char *name, int dflt) { return DBG_KERNELS ? env_int(name, dflt) : dflt; }

}  // namespace ak
