// placeholder: replaced below in this round
#include "common.h"
using namespace mb;
extern "C" int mb_taco_num_weights(const mb_taco_config*) { set_error("tacotron: not built yet"); return MB_ESTATE; }
extern "C" size_t mb_taco_weight_numel(const mb_taco_config*, int) { return 0; }
extern "C" int mb_taco_create(const mb_taco_config*, const float* const*, int, mb_taco**) { set_error("tacotron: not built yet"); return MB_ESTATE; }
extern "C" void mb_taco_destroy(mb_taco*) {}
extern "C" size_t mb_taco_workspace_bytes(const mb_taco*, int, int, int) { return 0; }
extern "C" int mb_taco_decode(const mb_taco*, const float*, const float*, const int32_t*, int, int, int, float,
                              const float*, uint64_t, float*, float*, float*, int*, void*, size_t, mb_stream_t) {
  set_error("tacotron: not built yet"); return MB_ESTATE;
}
