// Explicit instantiations of k_rowpass4, part 3 of 3 (rowpass4_list.hip.h): compiled in parallel with the api_*.hip units
// and linked into libklnmf.so by __graft_entry__.build() / scripts/build_lib.py.
#include "mfma4.hip.h"
#include "rowpass4_list.hip.h"

namespace klnmf {
KL_RP4_LIST_3(KL_RP4_DEFINE)
}  // namespace klnmf
