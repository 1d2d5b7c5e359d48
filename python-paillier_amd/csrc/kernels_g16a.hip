// kernels_g16a.hip — limb-group kernels for groups of 16 lanes, L in {1, 2, 3, 5, 7, 9} (see group_kernels.inc)
#define PHE_PART g16a
#define PHE_PART_G 16
#define PHE_FOR_EACH_L(X) X(1) X(2) X(3) X(5) X(7) X(9)
#include "group_kernels.inc"
