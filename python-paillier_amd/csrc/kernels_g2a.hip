// kernels_g2a.hip — limb-group kernels for groups of 2 lanes, L in {18} (see group_kernels.inc)
#define PHE_PART g2a
#define PHE_PART_G 2
#define PHE_FOR_EACH_L(X) X(18)
#include "group_kernels.inc"
