// mzx_fused_fc.h -- fused, LDS-resident whole-search kernel for fully connected
// networks (gfx950 only).  Placeholder until the kernel lands: reports
// "unsupported" so mzx_search_run uses the generic path.
#pragma once
#include "mzx_search.h"

namespace mzx {
inline int fused_fc_supported(const mzx_search*) { return 0; }
inline int fused_fc_run(mzx_search*, const mzx_search_io*, void*, stream_t) {
  set_error("fused search kernel not available");
  return MZX_ERR_INVALID;
}
}  // namespace mzx
