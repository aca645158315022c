// placeholder until the fused kernel lands
#include "pds_internal.h"
namespace pds {
int32_t fast_tables_create(pds_stft_plan *plan, const double *, const int32_t *, const int32_t *, const double *) { plan->fast.kind = 0; return PDS_OK; }
void fast_tables_destroy(pds_stft_plan *) {}
int32_t launch_stft_fast_f32(const pds_stft_plan *, const BatchArgs &) { set_error("fast kernel not built"); return PDS_ERR_INVALID; }
}
