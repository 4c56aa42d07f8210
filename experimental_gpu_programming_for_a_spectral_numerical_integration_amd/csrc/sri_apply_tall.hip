// sri_apply_tall.hip — every variant of sri::apply_kernel in the tall geometry (GeoTall: 256 x 128 tiles, 8 waves, one workgroup per CU: operators of more than 128 rows)
#include "sri_apply_geometry_impl.h"

SRI_DEFINE_GEOMETRY(tall, sri::GeoTall)
