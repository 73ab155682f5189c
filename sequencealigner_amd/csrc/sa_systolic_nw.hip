/* systolic kernels of one method: see sa_systolic_kernel.inc */
#include "sa_internal.h"
#define SA_SYS_METHOD SA_METHOD_NW
#define SA_SYS_LAUNCH sa_launch_systolic_nw
#define SA_SYS_WARM sa_warm_systolic_nw
#include "sa_systolic_kernel.inc"
