/*
 * rfx_abi.h -- the slice of RayforceDB's in-memory object ABI that the MI355X execution layer has to
 * agree on with the host evaluator.  Nothing here is code from the reference: it is a re-declaration
 * of layouts and constants, each with the reference location it must stay bit-compatible with
 * (paths relative to the RayforceDB tree).
 *
 *   object header ........ core/rayforce.h:112-133   (16 bytes, data at +16)
 *   type codes ........... core/rayforce.h:50-95
 *   null / inf sentinels . core/rayforce.h:97-107
 *   operator shapes ...... core/ops.h:199-204        (unary_f / binary_f / vary_f)
 *   function attributes .. core/ops.h:42-49
 *   group index object ... core/index.c:1696-1699    (7-slot list) + core/index.h (index types)
 *   column file .......... core/binary.c:263-311 writer, core/unary.c:48-136 reader
 */
#ifndef RFX_ABI_H
#define RFX_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- element type codes (vector = +code, atom = -code) -- core/rayforce.h:50-95 ---- */
enum {
    RFX_TYPE_LIST = 0,
    RFX_TYPE_B8 = 1,
    RFX_TYPE_U8 = 2,
    RFX_TYPE_I16 = 3,
    RFX_TYPE_I32 = 4,
    RFX_TYPE_I64 = 5,
    RFX_TYPE_SYMBOL = 6,
    RFX_TYPE_DATE = 7,
    RFX_TYPE_TIME = 8,
    RFX_TYPE_TIMESTAMP = 9,
    RFX_TYPE_F64 = 10,
    RFX_TYPE_ENUM = 20,      /* (key symbol, I64 indices) pair, or -- mmapped -- the indices with the key one page before -- core/util.h:103-105 */
    RFX_TYPE_MAPFILTER = 71, /* lazy (val, row-index) pair -- core/filter.c:29-49 */
    RFX_TYPE_MAPGROUP = 72,  /* lazy (val, group-index) pair -- core/group.c:26-46 */
    RFX_TYPE_MAPCOMMON = 74, /* virtual column of a parted table: (one value per partition, rows per partition) -- core/vary.c:354-361 */
    RFX_TYPE_PARTEDLIST = 77, /* + element type: LIST of one (mmapped) vector per partition -- core/rayforce.h:70-82 */
    RFX_TYPE_TABLE = 98,
    RFX_TYPE_DICT = 99,
    RFX_TYPE_UNARY = 101,
    RFX_TYPE_BINARY = 102,
    RFX_TYPE_VARY = 103,
    RFX_TYPE_NULL = 126,
    RFX_TYPE_ERR = 127
};

/* ---- sentinels -- core/rayforce.h:97-107 ---- */
#define RFX_NULL_I64 ((int64_t)0x8000000000000000LL)
#define RFX_INF_I64 ((int64_t)0x7FFFFFFFFFFFFFFFLL)
#define RFX_NULL_F64_BITS 0x7FF8000000000000ULL /* what `0 / 0.0` folds to with gcc/clang on x86-64 */
#define RFX_INF_F64_BITS 0x7FF0000000000000ULL

/* ---- function attributes -- core/ops.h:42-49 ---- */
enum { RFX_FN_NONE = 0, RFX_FN_ATOMIC = 4, RFX_FN_AGGR = 8, RFX_FN_SPECIAL_FORM = 16 };

/* ---- memory-model byte of a header -- core/heap.h:43 (0xfd = "external simple" = mmapped column file) ---- */
enum { RFX_MMOD_INTERNAL = 0xff, RFX_MMOD_EXTERNAL_SIMPLE = 0xfd };

/* ---- the 16-byte object header -- core/rayforce.h:112-133 ---- */
typedef struct rfx_obj {
    uint8_t mmod;  /* memory model / heap block flag */
    uint8_t order; /* heap block order */
    int8_t type;   /* >0 vector, <0 atom */
    uint8_t attrs; /* ATTR_* or, for function objects, FN_* */
    uint32_t rc;   /* reference count */
    union {
        int8_t b8;
        uint8_t u8;
        int16_t i16;
        int32_t i32;
        int64_t i64;
        double f64;
        struct rfx_obj *obj;
        struct {
            int64_t len;
            int8_t raw[]; /* column data starts here: header + 16 */
        };
    };
} rfx_obj_t, *rfx_obj_p;

#define RFX_AS_RAW(o) ((void *)((rfx_obj_p)(o) + 1))
#define RFX_AS_I64(o) ((int64_t *)RFX_AS_RAW(o))
#define RFX_AS_F64(o) ((double *)RFX_AS_RAW(o))
#define RFX_AS_B8(o) ((int8_t *)RFX_AS_RAW(o))
#define RFX_AS_LIST(o) ((rfx_obj_p *)RFX_AS_RAW(o))
#define RFX_IS_ERR(o) ((o)->type == RFX_TYPE_ERR)
#define RFX_IS_ATOM(o) ((o)->type < 0)

#ifndef __cplusplus
_Static_assert(sizeof(rfx_obj_t) == 16, "object header must be 16 bytes (core/rayforce.h:112-133)");
_Static_assert(offsetof(rfx_obj_t, type) == 2, "type byte at +2");
_Static_assert(offsetof(rfx_obj_t, rc) == 4, "refcount at +4");
_Static_assert(offsetof(rfx_obj_t, len) == 8, "vector length at +8");
_Static_assert(offsetof(rfx_obj_t, raw) == 16, "vector payload at +16");
#else
static_assert(sizeof(rfx_obj_t) == 16, "object header must be 16 bytes (core/rayforce.h:112-133)");
#endif

/* ---- operator shapes -- core/ops.h:202-204 ---- */
typedef rfx_obj_p (*rfx_unary_f)(rfx_obj_p);
typedef rfx_obj_p (*rfx_binary_f)(rfx_obj_p, rfx_obj_p);
typedef rfx_obj_p (*rfx_vary_f)(rfx_obj_p *, int64_t);

/* ---- group index object: LIST of 7 -- core/index.c:1696-1699, accessors :1650-1694 ----
 *   [0] i64 index type  [1] i64 group count  [2] group ids (I64 vec; per-row for IDS, per-slot for SHIFT)
 *   [3] i64 shift (= min key) or NULL_I64    [4] source key column (SHIFT only)   [5] filter ids or null
 *   [6] meta                                                                                         */
enum { RFX_INDEX_TYPE_IDS = 0, RFX_INDEX_TYPE_SHIFT = 1 };
#define RFX_INDEX_SCOPE_LIMIT (4096 * 128) /* core/index.h:29 -- dense table kept (SHIFT) up to this range */

/* ---- thread-pool policy constants that fix the CPU path's chunking -- core/pool.c:36-37 ---- */
#define RFX_PAGE_SIZE 4096
#define RFX_POOL_SPLIT_THRESHOLD (RFX_PAGE_SIZE * 4)
#define RFX_GROUP_MEMORY_BUDGET (64 * 1024 * 1024)

#ifdef __cplusplus
}
#endif
#endif /* RFX_ABI_H */
