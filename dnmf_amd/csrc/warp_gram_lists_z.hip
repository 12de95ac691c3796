// K3n, the Z >= 2 instantiations (warp_gram_lists.hip has the kernel and the reasons): a translation unit of their own,
// compiled with -fno-slp-vectorize (dnmf_amd/build.py: PER_FILE_FLAGS).
#define DNMF_K3N_TU_Z 1
#include "warp_gram_lists.hip"
