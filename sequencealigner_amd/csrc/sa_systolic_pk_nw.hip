/* packed-u16 systolic kernels of one method, 8-lane groups: see sa_systolic_pk.inc */
#include "sa_internal.h"
#define SA_SYS_METHOD SA_METHOD_NW
#define SA_PK_G 8
#define SA_SYS_LAUNCH sa_launch_systolic_pk_nw
#define SA_SYS_WARM sa_warm_systolic_pk_nw
#include "sa_systolic_pk.inc"
