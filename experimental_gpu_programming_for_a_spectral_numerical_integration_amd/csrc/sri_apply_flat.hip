// sri_apply_flat.hip — every variant of sri::apply_kernel in the flat geometry (GeoFlat: 64 x 64 tiles, 4 waves, three workgroups per CU: operators of at most 64 rows)
#include "sri_apply_geometry_impl.h"

SRI_DEFINE_GEOMETRY(flat, sri::GeoFlat)
