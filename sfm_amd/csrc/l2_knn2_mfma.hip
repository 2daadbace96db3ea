// placeholder: replaced by the fp32 MFMA kernel
#include "sfm_internal.h"
namespace sfm {
int l2_knn2_dev(sfm_ctx* ctx, int, const float*, int, size_t, const float*, int, size_t, int,
                int32_t*, float*) {
    return set_error(ctx, SFM_E_UNSUPPORTED, "L2 matching not built");
}
int l2_match_pairs_dev(sfm_ctx* ctx, int, const float*, size_t, int, const int32_t*, int,
                       const int32_t*, int, float, sfm_dmatch*, int, int32_t*) {
    return set_error(ctx, SFM_E_UNSUPPORTED, "L2 matching not built");
}
size_t l2_match_pairs_workspace(int, int, int) { return 0; }
}  // namespace sfm
