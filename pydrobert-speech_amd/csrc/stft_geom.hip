// One instantiated geometry of the fused STFT kernel (stft_wave_kernel.h) per object file: compiled with
// -DPDS_G_N1=.. -DPDS_G_N2=.. -DPDS_G_ROWS=.. -DPDS_G_MINW=.. for every PDS_GEOM line of stft_geoms.def (Makefile),
// so that the ~250 kernel instantiations of the library build in parallel.
#include "stft_wave_launch.h"

#define PDS_GEOM_NAME_(a, b, c) launch_geom_##a##_##b##_##c
#define PDS_GEOM_NAME(a, b, c) PDS_GEOM_NAME_(a, b, c)

namespace pds {

int32_t PDS_GEOM_NAME(PDS_G_N1, PDS_G_N2, PDS_G_ROWS)(const pds_stft_plan *plan, const BatchArgs &a) {
  return launch_wave<PDS_G_N1, PDS_G_N2, PDS_G_ROWS, PDS_G_MINW>(plan, a);
}

}  // namespace pds
