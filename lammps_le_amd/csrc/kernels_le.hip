// kernels_le.hip — USER-LE fixes on the device (placeholder launchers; implemented below in steps)
#include "device.h"
namespace lmp_le {
void launch_ex_load(DeviceState &, const ExLoadParams &, int) { throw LammpsError("fix ex_load: device path not built"); }
void launch_ex_unload(DeviceState &, const ExUnloadParams &, int) { throw LammpsError("fix ex_unload: device path not built"); }
void launch_extrusion(DeviceState &, const ExtrusionParams &, int) { throw LammpsError("fix extrusion: device path not built"); }
}
