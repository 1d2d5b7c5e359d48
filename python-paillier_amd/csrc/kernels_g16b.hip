// kernels_g16b.hip — limb-group kernels for groups of 16 lanes, L in {14, 18} (see group_kernels.inc)
#define PHE_PART g16b
#define PHE_PART_G 16
#define PHE_FOR_EACH_L(X) X(14) X(18)
#include "group_kernels.inc"
