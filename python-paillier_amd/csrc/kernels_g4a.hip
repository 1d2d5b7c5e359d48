// kernels_g4a.hip — limb-group kernels for groups of 4 lanes, L in {9, 18} (see group_kernels.inc)
#define PHE_PART g4a
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(9) X(18)
#include "group_kernels.inc"
