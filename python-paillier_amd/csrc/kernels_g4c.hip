// kernels_g4c.hip — limb-group kernels for groups of 4 lanes, L in {36} (see group_kernels.inc)
#define PHE_PART g4c
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(36)
#include "group_kernels.inc"
