/* TEST INFRASTRUCTURE ONLY — no-op stand-in for hwloc (absent in this image).  The reference's
 * worker pool (kt-kernel/cpu_backend/worker_pool.{h,cpp}) uses hwloc only to pin threads and bind
 * memory; arithmetic does not depend on it.  Every call succeeds and binds nothing. */
#ifndef KTX_ORACLE_HWLOC_SHIM_H
#define KTX_ORACLE_HWLOC_SHIM_H
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ktx_hwloc_bitmap_s { int dummy; }* hwloc_bitmap_t;
typedef hwloc_bitmap_t hwloc_cpuset_t;
typedef hwloc_bitmap_t hwloc_nodeset_t;
typedef const struct ktx_hwloc_bitmap_s* hwloc_const_bitmap_t;
struct hwloc_obj { hwloc_bitmap_t cpuset; hwloc_bitmap_t nodeset; unsigned os_index; unsigned logical_index; };
typedef struct hwloc_obj* hwloc_obj_t;
typedef struct ktx_hwloc_topology_s { int dummy; }* hwloc_topology_t;
typedef enum { HWLOC_OBJ_MACHINE, HWLOC_OBJ_PACKAGE, HWLOC_OBJ_CORE, HWLOC_OBJ_PU, HWLOC_OBJ_NUMANODE } hwloc_obj_type_t;
enum { HWLOC_CPUBIND_PROCESS = 1, HWLOC_CPUBIND_THREAD = 2, HWLOC_CPUBIND_STRICT = 4 };
typedef enum { HWLOC_MEMBIND_DEFAULT = 0, HWLOC_MEMBIND_FIRSTTOUCH = 1, HWLOC_MEMBIND_BIND = 2 } hwloc_membind_policy_t;
enum { HWLOC_MEMBIND_PROCESS = 1, HWLOC_MEMBIND_THREAD = 2, HWLOC_MEMBIND_STRICT = 4, HWLOC_MEMBIND_BYNODESET = 32 };
static struct ktx_hwloc_bitmap_s ktx_hwloc_bm_;
static struct hwloc_obj ktx_hwloc_obj_ = { &ktx_hwloc_bm_, &ktx_hwloc_bm_, 0, 0 };
static struct ktx_hwloc_topology_s ktx_hwloc_topo_;
static inline int hwloc_topology_init(hwloc_topology_t* t) { *t = &ktx_hwloc_topo_; return 0; }
static inline int hwloc_topology_load(hwloc_topology_t t) { (void)t; return 0; }
static inline void hwloc_topology_destroy(hwloc_topology_t t) { (void)t; }
static inline hwloc_obj_t hwloc_get_obj_by_type(hwloc_topology_t t, hwloc_obj_type_t ty, unsigned i) { (void)t; (void)ty; (void)i; return &ktx_hwloc_obj_; }
static inline hwloc_obj_t hwloc_get_obj_inside_cpuset_by_type(hwloc_topology_t t, hwloc_const_bitmap_t s, hwloc_obj_type_t ty, unsigned i) { (void)t; (void)s; (void)ty; (void)i; return &ktx_hwloc_obj_; }
static inline hwloc_bitmap_t hwloc_bitmap_alloc(void) { return (hwloc_bitmap_t)calloc(1, sizeof(struct ktx_hwloc_bitmap_s)); }
static inline void hwloc_bitmap_free(hwloc_bitmap_t b) { free(b); }
static inline int hwloc_bitmap_copy(hwloc_bitmap_t d, hwloc_const_bitmap_t s) { (void)d; (void)s; return 0; }
static inline int hwloc_bitmap_singlify(hwloc_bitmap_t b) { (void)b; return 0; }
static inline int hwloc_set_thread_cpubind(hwloc_topology_t t, unsigned long th, hwloc_const_bitmap_t s, int f) { (void)t; (void)th; (void)s; (void)f; return 0; }
static inline int hwloc_get_thread_cpubind(hwloc_topology_t t, unsigned long th, hwloc_bitmap_t s, int f) { (void)t; (void)th; (void)s; (void)f; return 0; }
static inline int hwloc_set_membind(hwloc_topology_t t, hwloc_const_bitmap_t s, hwloc_membind_policy_t p, int f) { (void)t; (void)s; (void)p; (void)f; return 0; }
#ifdef __cplusplus
}
#endif
#endif
