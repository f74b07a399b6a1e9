/* hqalloc.h -- C ABI of the worker-side resource allocator: which resource INDICES a task gets.
 *
 * Replaces, on a worker node, tako's `ResourceAllocator`
 *   /root/reference/crates/tako/src/internal/worker/resources/allocator.rs:25-236   (new, try_allocate, release_allocation, is_enabled)
 *   .../worker/resources/pool.rs:47-505                                             (index / group / sum pools, claim orders)
 *   .../worker/resources/concise.rs:21-203                                          (concise free state)
 *   .../worker/resources/groups.rs:61-155                                           (NUMA / coupling group solver, HiGHS there)
 * whose callers are the worker's reactor (worker/reactor.rs:60 try_allocate when a task may start, :94 release_allocation
 * when it ends, :277 is_enabled when choosing what to start).  The server-side tick (hqtick.h) only decides task -> worker and amounts; the indices
 * ("cpu 3, 7; gpu 1 with 0.5 share") are decided here, on the worker, from the state of what is running there.
 *
 * This library is host-only (libhqalloc.so, no HIP dependency): a worker node has no MI355X, and the allocator is a
 * sequential state machine over a few dozen indices.  It shares the exact MILP solver of the tick (csrc/milp.cpp).
 *
 * Conventions: amounts are `ResourceAmount` fixed point (units * 10000 + fractions, common/resources/amount.rs:7,26-36);
 * pools are indexed by `ResourceId`; resource names / labels are resolved by the caller (allocator.rs:38-55,
 * worker/resources/map.rs:15-58): a List pool's indices are 0..n-1, a Range pool's are start..=end, a Groups pool's are the
 * positions of its flattened groups.  One ctx per worker, single-threaded like the reference (`Rc`, no `Send`).
 * Functions return >= 0 on success and a negative HQALLOC_E_* code on error; nothing aborts the host.
 */
#ifndef HQALLOC_H
#define HQALLOC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HQALLOC_ABI_VERSION 1u
#define HQALLOC_FRACTIONS_PER_UNIT 10000u

/* ResourcePool variants (pool.rs:47-52). */
enum { HQALLOC_POOL_EMPTY = 0, HQALLOC_POOL_INDICES = 1, HQALLOC_POOL_GROUPS = 2, HQALLOC_POOL_SUM = 3 };

/* AllocationRequest variants (common/resources/request.rs:14-21). */
enum {
    HQALLOC_COMPACT = 0,
    HQALLOC_TIGHT = 1,
    HQALLOC_SCATTER = 2,
    HQALLOC_FORCE_COMPACT = 3,
    HQALLOC_FORCE_TIGHT = 4,
    HQALLOC_ALL = 5
};

enum {
    HQALLOC_E_INVALID = -1,   /* malformed argument (NULL, unknown kind, unsorted request, unknown allocation id) */
    HQALLOC_E_CAPACITY = -2,  /* caller-provided output arrays too small; nothing was claimed                     */
    HQALLOC_E_INTERNAL = -3   /* an invariant the reference asserts on was violated (hqalloc_last_error has it)   */
};

typedef struct hqalloc_ctx hqalloc_ctx;

/* `ResourceDescriptor` after name/label resolution (common/resources/descriptor.rs:295-298; allocator.rs:33-98).
 * Groups of pool r are group_off[r] .. group_off[r+1]-1 (an INDICES pool has exactly one, EMPTY and SUM pools none);
 * group g holds index[index_off[g] .. index_off[g+1]-1] in descriptor order (claims pop from the back, pool.rs:311-317). */
typedef struct {
    uint32_t abi_version;
    uint32_t n_resources;
    const uint8_t *pool_kind;    /* [n_resources] HQALLOC_POOL_*                       */
    const uint64_t *sum_size;    /* [n_resources] size of a SUM pool, ignored otherwise */
    const uint32_t *group_off;   /* [n_resources + 1]                                   */
    const uint32_t *index_off;   /* [n_groups_total + 1]                                */
    const uint32_t *index;       /* [n_indices_total] ResourceIndex values              */
    /* ResourceDescriptorCoupling (descriptor.rs:247-271), resource ids already mapped (allocator.rs:65-84) */
    uint32_t n_couplings;
    const uint32_t *coupling_resource1, *coupling_group1, *coupling_resource2, *coupling_group2;
    const uint16_t *coupling_weight;
} hqalloc_descriptor;

/* `ResourceRequest` entries, sorted by strictly increasing resource id (request.rs:204-208). */
typedef struct {
    uint32_t n_entries;
    const uint32_t *resource_id;
    const uint8_t *kind;     /* HQALLOC_COMPACT .. HQALLOC_ALL */
    const uint64_t *amount;  /* ignored for HQALLOC_ALL        */
} hqalloc_request;

/* `Allocation` (common/resources/allocation.rs:6-33), filled into caller-owned arrays.  Resource k of the allocation covers
 * index entries idx_off[k] .. idx_off[k+1]-1, in the reference's order (whole indices first, the fractional one last). */
typedef struct {
    uint64_t allocation_id;   /* out: handle for hqalloc_release                      */
    uint32_t cap_resources;   /* in: capacity of resource_id / amount (idx_off needs cap_resources + 1) */
    uint32_t cap_indices;     /* in: capacity of index / group_idx / fractions         */
    uint32_t n_resources;     /* out */
    uint32_t n_indices;       /* out */
    uint32_t *resource_id;    /* out [n_resources]                                     */
    uint64_t *amount;         /* out [n_resources]                                     */
    uint32_t *idx_off;        /* out [n_resources + 1]                                 */
    uint32_t *index;          /* out [n_indices] ResourceIndex                         */
    uint32_t *group_idx;      /* out [n_indices]                                       */
    uint32_t *fractions;      /* out [n_indices] 0 = whole index                       */
} hqalloc_allocation;

/* ResourceAllocator::new (allocator.rs:33-98). */
int hqalloc_create(const hqalloc_descriptor *desc, hqalloc_ctx **out_ctx);
void hqalloc_destroy(hqalloc_ctx *ctx);

/* ResourceAllocator::is_enabled (allocator.rs:206-213): 1 = the request could be allocated now, 0 = not. */
int hqalloc_is_enabled(hqalloc_ctx *ctx, const hqalloc_request *rq);

/* ResourceAllocator::try_allocate (allocator.rs:215-227): 1 = allocated (out filled), 0 = `None`.
 * HQALLOC_E_CAPACITY is detected before anything is claimed (sizes are known from the request). */
int hqalloc_try_allocate(hqalloc_ctx *ctx, const hqalloc_request *rq, hqalloc_allocation *out);

/* ResourceAllocator::release_allocation (allocator.rs:110-113). */
int hqalloc_release(hqalloc_ctx *ctx, uint64_t allocation_id);

/* State queries (what the reference's tests and `validate` look at):
 *   hqalloc_pool_free     pool.rs:555-566 `current_free`: whole free indices (or the free amount of a SUM pool)
 *   hqalloc_concise_sum   concise.rs:148-152 `amount_sum`; which = 0: of the live concise state (`free_resources`),
 *                         which = 1: of `pools[r].concise_state()`
 *   hqalloc_free_groups   the live concise state of resource r: units per group (returns the number of groups; writes at most cap)
 *   hqalloc_free_fractions  the fraction entries {index: fractions} of one group of that state, ascending index (returns the
 *                         number of entries; writes at most cap)
 *   hqalloc_validate      allocator.rs:229-235 + pool.rs:507-545: 0 = consistent */
int hqalloc_pool_free(const hqalloc_ctx *ctx, uint32_t resource, uint64_t *out_amount);
int hqalloc_concise_sum(const hqalloc_ctx *ctx, uint32_t resource, int which, uint64_t *out_amount);
int hqalloc_free_groups(const hqalloc_ctx *ctx, uint32_t resource, uint32_t cap, uint32_t *out_units, uint32_t *out_n_fraction_entries);
int hqalloc_free_fractions(const hqalloc_ctx *ctx, uint32_t resource, uint32_t group, uint32_t cap, uint32_t *out_index, uint32_t *out_fractions);
int hqalloc_validate(const hqalloc_ctx *ctx);

/* The reference's test helper `force_claim_from_groups` (worker/resources/test_allocator.rs:24-39): claims `amount` of a
 * GROUPS pool from the given groups with the Compact order, bypassing the group solver. */
int hqalloc_force_claim_from_groups(hqalloc_ctx *ctx, uint32_t resource, uint32_t n_groups, const uint32_t *groups, uint64_t amount,
                                    hqalloc_allocation *out);

const char *hqalloc_last_error(const hqalloc_ctx *ctx);
uint32_t hqalloc_abi_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
