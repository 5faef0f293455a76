#include "common.h"
#include "artspeech_hip.h"
// 8: ConvGemmArgs.slab_tr and as_conv_gemm_multi_post_f32's post_ln argument (round 5: assigned late), AsAdainArgs.col_w, as_forward_io's
// frame capacity, the host submissions and the debug checks of as_lanes (round 6)
extern "C" int as_abi_version(void) { return AS_ABI_VERSION; }
