// placeholder until the MFMA fast path lands
#include "dlm_internal.h"
namespace dlm {
bool mfma16_supported(const KArgs&) { return false; }
hipError_t launch_mfma16_filter(const KArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_mfma16_smoother(const KArgs&, hipStream_t) { return hipErrorNotSupported; }
}
