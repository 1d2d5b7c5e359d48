// kernels_g4b.hip — limb-group kernels for groups of 4 lanes, L in {27} (see group_kernels.inc)
#define PHE_PART g4b
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(27)
#include "group_kernels.inc"
