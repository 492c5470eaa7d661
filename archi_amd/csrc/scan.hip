// scan.hip -- MFMA candidate scan (fast path). Placeholder until the kernel lands.
#include "index.h"
namespace ak {
bool fast_supported(const Index &, int, int) { return false; }
FastPlan fast_plan(const Index &, int, int) { return FastPlan{0, 0, 0, 0, 0}; }
int fast_search(Index &, const float *, const float *, int, int, const uint8_t *, int64_t *, double *, int *, int *,
                int64_t *, void *, const FastPlan &, hipStream_t) {
    AK_FAIL(-7, "fast path not built");
}
}  // namespace ak
