// The packing instantiations of the 32-point-wave kernels (I2SDF_OPT_SAVES24: abars / gus / gas as packed 24-bit records, x3.h P24) as their own
// translation unit: mlp_x3.hip with I2SDF_X3_P24_TU defines only i2sdf_launch_igrad3_p24 / i2sdf_launch_sdf_bwd3_p24 (see the end of that file).
#define I2SDF_X3_P24_TU 1
#include "mlp_x3.hip"
