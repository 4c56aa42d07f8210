// sri_apply_mid.hip — every variant of sri::apply_kernel in the mid geometry (GeoMid: 128 x 128 tiles, 4 waves, two workgroups per CU: 65-128 rows, and larger operators whose last 256-row block would be at most half full)
#include "sri_apply_geometry_impl.h"

SRI_DEFINE_GEOMETRY(mid, sri::GeoMid)
