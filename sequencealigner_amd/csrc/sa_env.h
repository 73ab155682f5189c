/* sa_env.h -- every environment switch of libseqalign_hip.so in ONE table (README lists them).
 *
 * All of them are diagnostics, experiments or test hooks; none is needed to use the library.  The table is read
 * once per context (sa_ctx_create) and once per sa_hip_align / sa_hip_memory call -- never per launch -- so a test may
 * change a switch between two contexts of one process. */
#ifndef SA_ENV_H
#define SA_ENV_H

struct SaEnv {
	/* kernel-family selection (parity tests A/B the families against each other) */
	bool force_generic = false; /* SA_HIP_FORCE_GENERIC : pair-per-wave kernels for everything                     */
	bool no_pk = false;         /* SA_HIP_NO_PK         : no packed-u16 kernels (the s32 systolic classes run)      */
	bool no_pk16 = false;       /* SA_HIP_NO_PK16       : no 16-lane packed kernels                                 */
	bool no_sort = false;       /* SA_HIP_NO_SORT       : row streams in store order (no arranged copies)           */
	/* launch structure */
	bool concurrent_classes = false; /* SA_HIP_CONCURRENT_CLASSES : several launches of a range side by side on side streams
	                                  * (round 3's default; one after the other measured 4-8 % faster)        */
	bool one_tile_size = false;  /* SA_HIP_ONE_TILE_SIZE  : no small tiles at the end of a short launch             */
	int chunk = 0;               /* SA_HIP_CHUNK=n        : fixed row-stream length (1..32)                         */
	int pk_wgs = 0;              /* SA_HIP_PK_WGS=n       : persistent workgroups of a packed launch                */
	int stagger = 0;             /* SA_HIP_STAGGER=n      : start delay per wave slot (sleep periods)               */
	int rotate_prio = -1;        /* SA_HIP_ROTATE_PRIO=0/1: rotating wave priority (-1: by method)                   */
	int small_below = 16;        /* SA_HIP_SMALL_BELOW=n  : two tile sizes below n tiles per slot                   */
	int small_div = 4;           /* SA_HIP_SMALL_DIV=n    : small tiles = big / n                                   */
	int small_frac = 5;          /* SA_HIP_SMALL_FRAC=n   : 1/n of the range's pairs run in small tiles             */
	/* host delivery */
	bool no_pin = false;    /* SA_HIP_NO_PIN    : never page-lock the destination                                   */
	bool no_direct = false; /* SA_HIP_NO_DIRECT : batched copies instead of direct stores into a locked packed matrix */
	bool no_shells = false; /* SA_HIP_NO_SHELLS : host scatter for the full layout                                  */
	/* sa_hip_align */
	int devices = 0; /* SA_HIP_DEVICES=n : use the first n visible devices                                           */
	int split = 0;   /* SA_HIP_SPLIT=n   : n slices even when fewer devices are visible (testing aid)                */
	int gather = -1; /* SA_HIP_GATHER=0/1: multi-device path through dense shares + RCCL all-gather (-1: default)    */
	int tiles_split = 0; /* SA_HIP_TILES_SPLIT=n: sa_hip_tiles_begin deals the column blocks over n jobs on device 0 (testing aid) */
	/* diagnostics */
	bool verbose = false;             /* SA_HIP_VERBOSE                                                               */
	bool stamps = false;              /* SA_HIP_STAMPS : per-tile clocks of every launch (synchronous)                */
	const char *stamps_dump = nullptr; /* SA_HIP_STAMPS_DUMP=<file> : raw stamp words                                 */
	bool ztrace = false;              /* SA_HIP_ZTRACE : the tile walk (sa_deflate.hip) prints what every block / batch cost  */
};

/* a fresh snapshot of the process environment */
SaEnv sa_env_read();

#endif /* SA_ENV_H */
