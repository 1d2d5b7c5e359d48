// kernels_g8b.hip — limb-group kernels for groups of 8 lanes, L in {18} (see group_kernels.inc)
#define PHE_PART g8b
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(18)
#include "group_kernels.inc"
