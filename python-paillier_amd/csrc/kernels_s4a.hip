// kernels_s4a.hip — split-modulus kernels for groups of 4 lanes, L in {9, 14} (see split_kernels.inc)
#define PHE_PART s4a
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(9) X(14)
#include "split_kernels.inc"
