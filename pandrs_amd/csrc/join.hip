// join.hip — placeholder until the radix-partitioned hash join lands.
#include "common.hpp"
namespace pandrs {
int32_t join_entry(pandrs_hip_ctx *, int32_t, const pandrs_hip_column *, int64_t,
                   const pandrs_hip_column *, int64_t, int32_t, int64_t *) {
    return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "join: not implemented yet");
}
int32_t join_groupby_sum_entry(pandrs_hip_ctx *, int32_t, const pandrs_hip_column *,
                               const pandrs_hip_column *, int64_t, const pandrs_hip_column *,
                               const pandrs_hip_column *, int64_t, int64_t *) {
    return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "join_groupby_sum: not implemented yet");
}
}  // namespace pandrs
