// Fused red+black sweep (one launch, one read and one write of u per sweep).
// Placeholder until the z-streaming kernel lands: reports "not handled" so the
// two-pass colour kernels of smooth.hip run.
#include "common.hpp"

namespace ndsm {

int launch_rbgs3_fused(const ndsmk_grid &g, double *u, const double *rhs, bool *handled) {
  (void)g;
  (void)u;
  (void)rhs;
  *handled = false;
  return 0;
}

}  // namespace ndsm
