// kernels_s8b.hip — split-modulus kernels for groups of 8 lanes, L in {14} (see split_kernels.inc)
#define PHE_PART s8b
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(14)
#include "split_kernels.inc"
