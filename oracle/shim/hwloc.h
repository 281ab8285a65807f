/* TEST INFRASTRUCTURE ONLY — a small stand-in for hwloc (absent in this image), enough for what the reference's worker pool
 * (kt-kernel/cpu_backend/worker_pool.{h,cpp}:43-92, :269-330) asks of it: the CPUs of a NUMA node, the i-th physical core inside
 * them, and binding a thread to one hardware thread of that core.  Round 4: the first version of this file
 * bound nothing, so bench.py's cpu_baseline ran the reference's kernels on unpinned threads; now the topology is read from sysfs
 * (/sys/devices/system/node/nodeN/cpulist, cpuN/topology/thread_siblings_list) and hwloc_set_thread_cpubind is
 * pthread_setaffinity_np — the placement the reference's README asks for (one worker per physical core, pinned).  Memory binding
 * stays with the reference's own libnuma calls.  Arithmetic does not depend on any of this. */
#ifndef KTX_ORACLE_HWLOC_SHIM_H
#define KTX_ORACLE_HWLOC_SHIM_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef __cplusplus
extern "C" {
#endif
#define KTX_HWLOC_MAXCPU 2048
typedef struct ktx_hwloc_bitmap_s { unsigned long w[KTX_HWLOC_MAXCPU / (8 * sizeof(unsigned long))]; }* hwloc_bitmap_t;
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

#define KTX_HWLOC_BITS (8 * (int)sizeof(unsigned long))
static inline int ktx_hwloc_isset(hwloc_const_bitmap_t b, int i) { return (int)((b->w[i / KTX_HWLOC_BITS] >> (i % KTX_HWLOC_BITS)) & 1ul); }
static inline void ktx_hwloc_set(hwloc_bitmap_t b, int i) { b->w[i / KTX_HWLOC_BITS] |= 1ul << (i % KTX_HWLOC_BITS); }
/* "0-63,128-191" -> bits */
static inline int ktx_hwloc_parse_list(const char* path, hwloc_bitmap_t out) {
  memset(out, 0, sizeof(*out));
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  int any = 0;
  for (char* p = buf; *p;) {
    char* e;
    long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
    for (long i = a; i <= b && i < KTX_HWLOC_MAXCPU; i++) { ktx_hwloc_set(out, (int)i); any = 1; }
    p = e;
    while (*p == ',' || *p == '\n' || *p == ' ') p++;
  }
  return any ? 0 : -1;
}

static struct ktx_hwloc_topology_s ktx_hwloc_topo_;
static inline int hwloc_topology_init(hwloc_topology_t* t) { *t = &ktx_hwloc_topo_; return 0; }
static inline int hwloc_topology_load(hwloc_topology_t t) { (void)t; return 0; }
static inline void hwloc_topology_destroy(hwloc_topology_t t) { (void)t; }
static inline hwloc_bitmap_t hwloc_bitmap_alloc(void) { return (hwloc_bitmap_t)calloc(1, sizeof(struct ktx_hwloc_bitmap_s)); }
static inline void hwloc_bitmap_free(hwloc_bitmap_t b) { free(b); }
static inline int hwloc_bitmap_copy(hwloc_bitmap_t d, hwloc_const_bitmap_t s) { *d = *s; return 0; }
static inline int hwloc_bitmap_singlify(hwloc_bitmap_t b) {   /* keep the lowest set bit only */
  int first = -1;
  for (int i = 0; i < KTX_HWLOC_MAXCPU && first < 0; i++)
    if (ktx_hwloc_isset(b, i)) first = i;
  memset(b, 0, sizeof(*b));
  if (first >= 0) ktx_hwloc_set(b, first);
  return 0;
}
/* objects are leaked on purpose (a handful per pool, test infrastructure) so that the pointers the caller keeps stay valid */
static inline hwloc_obj_t ktx_hwloc_new_obj(unsigned idx) {
  hwloc_obj_t o = (hwloc_obj_t)calloc(1, sizeof(struct hwloc_obj));
  o->cpuset = hwloc_bitmap_alloc();
  o->nodeset = hwloc_bitmap_alloc();
  o->os_index = o->logical_index = idx;
  return o;
}
static inline hwloc_obj_t hwloc_get_obj_by_type(hwloc_topology_t t, hwloc_obj_type_t ty, unsigned i) {
  (void)t;
  hwloc_obj_t o = ktx_hwloc_new_obj(i);
  char path[128];
  if (ty == HWLOC_OBJ_NUMANODE) {
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%u/cpulist", i);
    if (ktx_hwloc_parse_list(path, o->cpuset) != 0) {   /* no such node (or no sysfs): node 0 = every online CPU, others absent */
      if (i != 0 || ktx_hwloc_parse_list("/sys/devices/system/cpu/online", o->cpuset) != 0) return NULL;
    }
    ktx_hwloc_set(o->nodeset, (int)i);
    return o;
  }
  if (ktx_hwloc_parse_list("/sys/devices/system/cpu/online", o->cpuset) != 0) return NULL;
  return o;
}
/* the idx-th physical core inside `s`: cores in ascending order of their lowest hardware thread */
static inline hwloc_obj_t hwloc_get_obj_inside_cpuset_by_type(hwloc_topology_t t, hwloc_const_bitmap_t s, hwloc_obj_type_t ty, unsigned idx) {
  (void)t; (void)ty;
  unsigned seen = 0;
  struct ktx_hwloc_bitmap_s sib;
  for (int c = 0; c < KTX_HWLOC_MAXCPU; c++) {
    if (!ktx_hwloc_isset(s, c)) continue;
    char path[160];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    if (ktx_hwloc_parse_list(path, &sib) != 0) { memset(&sib, 0, sizeof(sib)); ktx_hwloc_set(&sib, c); }
    int lowest = -1;
    for (int i = 0; i < KTX_HWLOC_MAXCPU && lowest < 0; i++)
      if (ktx_hwloc_isset(&sib, i)) lowest = i;
    if (lowest != c) continue;        /* counted with its lowest hardware thread */
    if (seen++ == idx) {
      hwloc_obj_t o = ktx_hwloc_new_obj(idx);
      *o->cpuset = sib;
      return o;
    }
  }
  return NULL;
}
static inline int hwloc_set_thread_cpubind(hwloc_topology_t t, pthread_t th, hwloc_const_bitmap_t s, int f) {
  (void)t; (void)f;
  const char* nobind = getenv("KTX_HWLOC_NOBIND");       /* bench.py's placement sweep: the same pool, threads left to the scheduler */
  if (nobind && nobind[0] == '1') return 0;
  cpu_set_t* cs = CPU_ALLOC(KTX_HWLOC_MAXCPU);
  if (!cs) return -1;
  const size_t sz = CPU_ALLOC_SIZE(KTX_HWLOC_MAXCPU);
  CPU_ZERO_S(sz, cs);
  int any = 0;
  for (int i = 0; i < KTX_HWLOC_MAXCPU; i++)
    if (ktx_hwloc_isset(s, i)) { CPU_SET_S(i, sz, cs); any = 1; }
  const int rc = any ? pthread_setaffinity_np(th, sz, cs) : 0;
  CPU_FREE(cs);
  return rc == 0 ? 0 : -1;     /* (a container that forbids the CPU: the thread stays unpinned, the pool still works) */
}
static inline int hwloc_get_thread_cpubind(hwloc_topology_t t, pthread_t th, hwloc_bitmap_t s, int f) {
  (void)t; (void)f;
  cpu_set_t* cs = CPU_ALLOC(KTX_HWLOC_MAXCPU);
  if (!cs) return -1;
  const size_t sz = CPU_ALLOC_SIZE(KTX_HWLOC_MAXCPU);
  memset(s, 0, sizeof(*s));
  const int rc = pthread_getaffinity_np(th, sz, cs);
  if (rc == 0)
    for (int i = 0; i < KTX_HWLOC_MAXCPU; i++)
      if (CPU_ISSET_S(i, sz, cs)) ktx_hwloc_set(s, i);
  CPU_FREE(cs);
  return rc == 0 ? 0 : -1;
}
static inline int hwloc_set_membind(hwloc_topology_t t, hwloc_const_bitmap_t s, hwloc_membind_policy_t p, int f) { (void)t; (void)s; (void)p; (void)f; return 0; }
#define hwloc_bitmap_foreach_begin(id, bitmap) do { for ((id) = 0; (id) < KTX_HWLOC_MAXCPU; (id)++) { if (!ktx_hwloc_isset((bitmap), (int)(id))) continue;
#define hwloc_bitmap_foreach_end() } } while (0)
#ifdef __cplusplus
}
#endif
#endif
