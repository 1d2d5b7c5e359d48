// kernels_g8a.hip — limb-group kernels for groups of 8 lanes, L in {5, 9, 14} (see group_kernels.inc)
#define PHE_PART g8a
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(5) X(9) X(14)
#include "group_kernels.inc"
