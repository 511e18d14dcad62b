// Entry points declared in include/wmd.h whose kernels are not written yet.  They fail loudly
// (WMD_ERR_UNSUPPORTED + message) — there is no CPU fallback anywhere in this library.
#include "wmd_internal.h"
using namespace wmd;

#define WMD_TODO(name) return fail(WMD_ERR_UNSUPPORTED, #name ": not implemented yet")

extern "C" int wmd_minmax(const float*, size_t, float*, void*, size_t, void*) { WMD_TODO(wmd_minmax); }
extern "C" int wmd_mask_threshold(const float*, const float*, float, int, uint8_t*, int, int, void*) { WMD_TODO(wmd_mask_threshold); }
extern "C" int wmd_mask_dilate(const uint8_t*, uint8_t*, int, int, int, int, void*) { WMD_TODO(wmd_mask_dilate); }
extern "C" size_t wmd_mask_compact_workspace_bytes(int) { return 0; }
extern "C" int wmd_mask_compact(const uint8_t*, int32_t*, int32_t*, int32_t*, int, void*, size_t, void*) { WMD_TODO(wmd_mask_compact); }
extern "C" int wmd_sparse_conv(const wmd_sparse_conv_args*, void*) { WMD_TODO(wmd_sparse_conv); }
extern "C" int wmd_comm_unique_id(void*) { WMD_TODO(wmd_comm_unique_id); }
extern "C" int wmd_comm_init(wmd_comm**, const void*, int, int) { WMD_TODO(wmd_comm_init); }
extern "C" int wmd_comm_allreduce(wmd_comm*, float*, size_t, float, void*) { WMD_TODO(wmd_comm_allreduce); }
extern "C" int wmd_comm_destroy(wmd_comm*) { WMD_TODO(wmd_comm_destroy); }
