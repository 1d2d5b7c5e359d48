// kernels_g2b.hip — limb-group kernels for groups of 2 lanes, L in {36} (see group_kernels.inc)
#define PHE_PART g2b
#define PHE_PART_G 2
#define PHE_FOR_EACH_L(X) X(36)
#include "group_kernels.inc"
