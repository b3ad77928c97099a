/* invpref_plan.h -- C ABI of the host-side row-plan builder (libinvpref_ingest.so; plain C, no GPU involved).
 *
 * The planned M-step (include/invpref_hip.h: InvPrefRowPlan) needs the scatter pattern of every minibatch inverted
 * once.  The reference's minibatches are static (utils.mini_batch, utils.py:12-19: contiguous, unshuffled slices of the
 * resident interaction tensors), so this is set-up work, but it stands in front of the first planned step: 31 plans of a
 * Yahoo-shaped run took 0.35 s in numpy and a 2^24-interaction plan minutes.  This builder produces THE SAME arrays as
 * invpref_kdd_2022_amd/plan.py's numpy reference implementation, byte for byte (tests/test_plan_native.py), from the
 * same resolved parameters: two stable counting sorts, one pass per XCD class and side, no Python in the loop.
 *
 * Returns an opaque handle; the arrays are read out with invpref_plan_array() (int32 each, owned by the handle) and the
 * handle is released with invpref_plan_free().  invpref_plan_build_many() builds the plans of many minibatches
 * (consecutive slices of the same interaction arrays) on a thread pool. */
#ifndef INVPREF_PLAN_H
#define INVPREF_PLAN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct InvPrefPlanParams {
    int32_t lanes_per_group;          /* 16 / 32: NG = 256 / lanes_per_group group slots per round */
    int32_t per_slice, item_per_slice;
    int32_t rounds_per_task, item_rounds_per_task;
    int32_t n_classes;                /* 1 .. 8 */
    int32_t rows_per_stream_task;
    int32_t push;                     /* item side in push form: no inline interactions (rec_slot serves as push_slot) */
    int32_t user_lo, user_hi;         /* untouched user rows outside [user_lo, user_hi) are not streamed; (0, user_num) = all */
    int32_t fill_cap;                 /* launch-1 residency the stream split fills (0: plain split) */
    int32_t snake_user;               /* > 0: launch 1's rounds of a class heaviest first, every other row of this many reversed
                                       * (tasks j, j + 32 .. of an XCD share a CU: heavy rounds meet light ones) */
    double stream_split;              /* share of a class's untouched rows streamed by launch 1 */
} InvPrefPlanParams;

typedef struct InvPrefHostPlan InvPrefHostPlan;

/* which: 0 user_desc [rounds][NG][8] | 1 item_desc | 2 user_round_iters | 3 user_list [n][4] = (item, position, label
 *        bits, slot) | 4 item_list [n][2] = (user, slot) | 5 stream_rows | 6 rec_slot [n] (position -> slot = index in the
 *        item order; InvPrefRowPlan.rec_slot, and .push_slot in push form) | 7 cls [8][8] */
InvPrefHostPlan *invpref_plan_build(const int64_t *users, const int64_t *items, const float *scores, int64_t n,
                                    int64_t user_num, int64_t item_num, const InvPrefPlanParams *params);
int64_t invpref_plan_array(const InvPrefHostPlan *plan, int32_t which, const int32_t **data);
void invpref_plan_free(InvPrefHostPlan *plan);

/* interactions per row of one side (what np.bincount(rows, minlength=n_rows) returns), on threads: the plan's parameters
 * (slice lengths, rounds per task) are chosen from these before the arrays are built.  0, -1 bad arguments, -2 a row id
 * outside [0, n_rows). */
int invpref_plan_row_counts(const int64_t *rows, int64_t n, int64_t n_rows, int64_t *counts);

/* plans of `count` minibatches: minibatch k = interactions [offsets[k], offsets[k + 1]) with parameters params[k];
 * out[k] receives its handle (NULL on failure).  n_threads <= 0: one per hardware thread (at most 32). */
int invpref_plan_build_many(const int64_t *users, const int64_t *items, const float *scores, const int64_t *offsets,
                            int32_t count, int64_t user_num, int64_t item_num, const InvPrefPlanParams *params,
                            InvPrefHostPlan **out, int32_t n_threads);

/* ---- alt plans (include/invpref_hip.h: InvPrefAltPlan; plan.py: build_alt_plan is the numpy reference implementation the
 * arrays equal byte for byte, tests/test_plan_native.py).  One launch of the alternating form: the CURRENT minibatch
 * (cur_* arrays, n interactions; n = 0: a flush launch) seen from `side` (0: users own the jobs, 1: items) and the PREVIOUS
 * minibatch (prev_*, n_prev interactions; n_prev = 0 with prev_users == NULL: first launch of a run).
 * which: 0 desc [rounds][slots][8] | 1 pend [rounds][slots][4] | 2 list [n][4] | 3 push_slot [n] | 4 stream [n_stream][4] |
 *        5 cls [8][4]. */
typedef struct InvPrefAltPlanParams {
    int32_t side, per_slice, n_classes;
    int32_t pend_job_min;             /* rows without a current interaction and at least this many pending rows get a job */
    int32_t pend_per_slice;
    int32_t slots;                    /* group slots per round: 16 or 32 (InvPrefAltPlan.slots_per_round) */
} InvPrefAltPlanParams;
InvPrefHostPlan *invpref_alt_plan_build(const int64_t *cur_users, const int64_t *cur_items, const float *cur_scores, int64_t n,
                                        const int64_t *prev_users, const int64_t *prev_items, int64_t n_prev,
                                        int64_t user_num, int64_t item_num, const InvPrefAltPlanParams *params);
/* `count` alt plans over the same interaction arrays: plan k evaluates interactions [cur_lo[k], cur_lo[k] + cur_n[k]) (cur_n
 * 0: a flush) after a launch that evaluated [prev_lo[k], prev_lo[k] + prev_n[k]) (prev_n[k] < 0: none), from side[k]; the
 * other parameters are shared.  out[k] receives the handle.  n_threads <= 0: one per hardware thread (at most 32). */
int invpref_alt_plan_build_many(const int64_t *users, const int64_t *items, const float *scores, const int64_t *cur_lo,
                                const int64_t *cur_n, const int64_t *prev_lo, const int64_t *prev_n, const int32_t *side,
                                int32_t count, int64_t user_num, int64_t item_num, const InvPrefAltPlanParams *params,
                                InvPrefHostPlan **out, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
