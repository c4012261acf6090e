// Level-2 rule automaton, LDS tier: the source of l2_kernel.hip compiled with the per-document hot
// state in a 77 KB LDS slice per one-wave workgroup (see "where the per-document state lives" there).
#define SPA_L2_LDS 1
#include "l2_kernel.hip"
