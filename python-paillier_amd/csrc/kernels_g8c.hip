// kernels_g8c.hip — limb-group kernels for groups of 8 lanes, L in {27} (see group_kernels.inc)
#define PHE_PART g8c
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(27)
#include "group_kernels.inc"
