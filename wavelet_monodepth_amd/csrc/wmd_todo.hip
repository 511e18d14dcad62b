// Entry points declared in include/wmd.h whose kernels are not written yet.  They fail loudly
// (WMD_ERR_UNSUPPORTED + message) — there is no CPU fallback anywhere in this library.
#include "wmd_internal.h"
using namespace wmd;

#define WMD_TODO(name) return fail(WMD_ERR_UNSUPPORTED, #name ": not implemented yet")

extern "C" int wmd_comm_unique_id(void*) { WMD_TODO(wmd_comm_unique_id); }
extern "C" int wmd_comm_init(wmd_comm**, const void*, int, int) { WMD_TODO(wmd_comm_init); }
extern "C" int wmd_comm_allreduce(wmd_comm*, float*, size_t, float, void*) { WMD_TODO(wmd_comm_allreduce); }
extern "C" int wmd_comm_destroy(wmd_comm*) { WMD_TODO(wmd_comm_destroy); }
