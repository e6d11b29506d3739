// placeholder until the fused inference kernels land (same session)
#include "mz_common.h"
struct mz_net { int kind; };
extern "C" {
int64_t mz_net_num_weights(int32_t) { return 0; }
int32_t mz_net_hidden_size(int32_t) { return 0; }
int mz_net_create(int32_t, const float*, int64_t, int64_t, void*, mz_net_t**) { return mz::fail(MZ_ERR_STATE, "networks not built yet"); }
int mz_net_destroy(mz_net_t*) { return MZ_OK; }
int mz_net_set_stream(mz_net_t*, void*) { return MZ_OK; }
int64_t mz_net_training_steps(mz_net_t*) { return 0; }
int mz_infer_initial(mz_net_t*, int32_t, const float*, float*, float*, float*, float*) { return mz::fail(MZ_ERR_STATE, "networks not built yet"); }
int mz_infer_recurrent(mz_net_t*, int32_t, const float*, const int32_t*, int64_t, const int32_t*, float*, float*, float*, float*, float*) { return mz::fail(MZ_ERR_STATE, "networks not built yet"); }
}
