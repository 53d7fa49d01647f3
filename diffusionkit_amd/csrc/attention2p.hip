// attention2.hip compiled a second time under LLVM's default machine scheduler: the D = 128 instantiations of dk_attn2_fwd_kernel
// (see the end of attention2.hip and the Makefile).
#define DK_ATTN2_DEFAULT_SCHED_TU 1
#include "attention2.hip"
