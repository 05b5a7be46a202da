// k_poa_mw.hip -- the LAST pass of k_poa as a workgroup of eight waves per read (see the C3_POA_MW sections of k_poa.hip): the
// reads that end up here are the ones whose band blew up to the width of the subread, and one wave takes hundreds of
// milliseconds for each.  The same source as the single-wave kernels, compiled a second time in a namespace of its own.
#include "c3_dev.h"
#include "c3_args.h"
#include <type_traits>
#ifndef C3_POA_MW
#define C3_POA_MW 8
#endif
namespace c3mw {
#include "k_poa.hip"
}
