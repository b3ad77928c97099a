/*
 * invpref_hip.h -- C ABI of the MI355X-native InvPref hot path (libinvpref_hip.so).
 *
 * The reference (AIflowerQ/InvPref_KDD_2022) is pure Python/PyTorch and has no FFI boundary; its
 * seam for this path is the Python object surface of models.py / train.py.  Each entry point below
 * names the reference code it replaces (file:line relative to the reference root).  All pointers
 * are DEVICE pointers unless stated otherwise; the caller owns every buffer; nothing is retained
 * after a call returns; every launch is enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = default stream) and no call synchronises the host
 * unless stated.  Return value: 0 on success, a negative INVPREF_E* code for bad arguments, or a
 * positive hipError_t passed through.  No exceptions cross this boundary.
 *
 * Tables are fp32 row-major; ids are int64 (the reference uses torch.LongTensor, train.py:31-34).
 * Out-of-range ids are undefined behaviour on the device (the reference raises IndexError).
 */
#ifndef INVPREF_HIP_H
#define INVPREF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INVPREF_ABI_VERSION 6

/* error codes */
#define INVPREF_OK 0
#define INVPREF_EINVAL (-1)      /* null pointer / negative size */
#define INVPREF_EUNSUPPORTED (-2) /* factor_num or env_num outside the compiled range */
#define INVPREF_EWORKSPACE (-3)  /* workspace too small */

/* limits of the compiled kernels */
#define INVPREF_MAX_FACTORS 256
#define INVPREF_MAX_ENVS 16

/* flags (bit set) */
#define INVPREF_IMPLICIT 1u       /* InvPrefImplicit + BCELoss (models.py:272, train.py:42); else explicit + MSELoss (models.py:414, train.py:719) */
#define INVPREF_REWEIGHT_REC 2u   /* use_recommend_re_weight  train.py:125-142 */
#define INVPREF_REWEIGHT_CLS 4u   /* use_class_re_weight      train.py:120-136 */
#define INVPREF_REG_ONLY_EMBED 8u /* model.reg_only_embed     models.py:369,381 */
#define INVPREF_REG_ENV_EMBED 16u /* model.reg_env_embed      models.py:376,388 */
#define INVPREF_DENSE_REG 32u     /* add the classifier regulariser (models.py:211-217) in this call;
                                     set on exactly one rank when a minibatch is row-sharded */

#define INVPREF_NO_GRAD 64u       /* invpref_mstep_grad_hip: report the loss terms only, write no gradient */
#define INVPREF_PURE_MF 128u      /* planned M-step entry points (invpref_mstep_rows_*): PureMF baseline step
                                     (baseline_models.py:12-69, :652-704 under Basic*TrainManager, train.py:345-461,
                                     :1022-1138) = the InvPref step with the env-aware tables, embed_env and the
                                     classifier ABSENT: env_num must be 1, those five pointers (and `envs`) are
                                     ignored and may be NULL in every table struct.  With coefficients
                                     (1, 0, 0, 2*L2_coe, 2*L1_coe, 0) the reported terms are
                                     losses6 = {score_loss, -, 0, L2_reg/2, L1_reg/2, loss} of train.py:399-404 */
#define INVPREF_WEIGHTS_BY_ENV 256u /* planned M-step entry points (invpref_mstep_rows_*, invpref_mstep_alt_hip): the
                                     `sample_weights` argument holds class_weights[env_num] -- the E floats stat_envs forms
                                     (train.py:274-277) -- and an interaction's weight is class_weights[envs[i]], which IS
                                     sample_weights[i] (train.py:278) as long as stat_envs() has run since the environments
                                     last changed (train.py:329-330 calls them together).  Staged with the small tables: no
                                     per-interaction weight load.  Without the flag sample_weights[i] is read as before
                                     (train_a_batch on caller tensors, train.py:94-106). */

/* The seven parameter tensors of InvPrefImplicit/InvPrefExplicit (models.py:283-291, :197-201),
 * in state_dict order.  Also used for gradients and Adam moments (same shapes). */
typedef struct InvPrefTables {
    int64_t user_num, item_num, env_num, factor_num;
    float *embed_user_invariant;   /* [user_num, factor_num] */
    float *embed_item_invariant;   /* [item_num, factor_num] */
    float *embed_user_env_aware;   /* [user_num, factor_num] */
    float *embed_item_env_aware;   /* [item_num, factor_num] */
    float *embed_env;              /* [env_num,  factor_num] */
    float *classifier_weight;      /* [env_num,  factor_num]  env_classifier.linear_map.weight */
    float *classifier_bias;        /* [env_num]               env_classifier.linear_map.bias   */
} InvPrefTables;

/* loss coefficients of ImplicitTrainManager/ExplicitTrainManager (train.py:22, :45-49) + the
 * gradient-reversal alpha handed to forward() (train.py:108, functions.py:7-16) */
typedef struct InvPrefCoefs {
    float invariant_coe, env_aware_coe, env_coe, L2_coe, L1_coe, alpha;
} InvPrefCoefs;

int invpref_abi_version(void);
/* name of the device the library would launch on (host buffer), for bench/logging */
int invpref_device_name(char *buf, size_t len);

/* ---- forward: replaces InvPref{Implicit,Explicit}.forward (models.py:307-326, :448-467).
 * out: invariant_score[B], env_aware_score[B], env_outputs[B, env_num] (log-softmax). */
int invpref_forward_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items,
                        const int64_t *envs, int64_t B, uint32_t flags, float *invariant_score,
                        float *env_aware_score, float *env_outputs, void *stream);

/* ---- M-step gradient: replaces the forward + 3 losses + 2 regularisers + loss.backward() of
 * train_a_batch (train.py:94-156; models.py:307-391; functions.py:4-16).
 * Gradients are ADDED into `grads` (zero it first: invpref_adam_hip(..., zero_grad=1) leaves it
 * zeroed).  `batch_norm` is the mean() denominator -- the length of the whole minibatch, which is
 * larger than B when the minibatch is row-sharded over several GPUs.  losses6 (device, fp32[6]) is
 * ADDED to: {invariant_loss, env_aware_loss, envs_loss, L2_reg, L1_reg, loss} as in train.py:159-166
 * (zero it first; `loss` is the coefficient-weighted sum of this call's five partial terms, so it
 * is additive across shards as well).  sample_weights may be NULL when no re-weight flag is set.
 * workspace: device scratch of at least invpref_mstep_workspace_bytes() bytes. */
size_t invpref_mstep_workspace_bytes(const InvPrefTables *tables, int64_t B);
int invpref_mstep_grad_hip(const InvPrefTables *tables, const InvPrefTables *grads, const int64_t *users,
                           const int64_t *items, const int64_t *envs, const float *scores,
                           const float *sample_weights, int64_t B, int64_t batch_norm,
                           const InvPrefCoefs *coefs, uint32_t flags, float *losses6, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- planned, atomic-free M-step ("row jobs"), two launches.  utils.mini_batch (utils.py:12-19) yields the same
 * contiguous, unshuffled slices every epoch, so the scatter pattern of a minibatch is inverted once into a plan
 * (built on the host, invpref_kdd_2022_amd/plan.py).  All arrays are device memory and are only read.
 *   group = the lanes that hold one embedding row: lanes_per_group = 16 / 16 / 32 for factor_num <= 64 / 128 / 256
 *           (invpref_rows_lanes_per_group()) with 1 / 2 / 2 float4 per lane; NG = 256 / lanes_per_group groups per
 *           workgroup;
 *   job   = one row of the user tables or of the item tables + the minibatch's interactions that touch it,
 *           cut into 1, 2, 4, ... NG equal slices (a power of two), one slice per group;
 *   round = the NG group slots of one workgroup; all jobs of a round have the same slice count; slices of a
 *           job sit in adjacent slots, the first one is the leader;
 *   task  = *_rounds_per_task consecutive rounds of one side, run by one workgroup.
 * Launch 1 (mstep_eval_kernel) runs the USER jobs: every interaction is evaluated there, once, and leaves a
 * record {g_p, g_q, env, gz[0..E)} in the workspace; launch 2 (mstep_apply_kernel) runs the ITEM jobs on those
 * records and folds the per-workgroup partial sums of launch 1 into embed_env / classifier / the loss outputs.
 * desc[round][slot] is 8 int32:
 *   {row (-1: idle slot), meta, a, b, c, d, e, f}
 *   meta = leader | slices << 1 | mode << 6 | row_count << 9       (idle slots carry slices too)
 *   user side: mode 0 / 1 / 2 = that many interactions inline as (item row, position, label bits) = (a, b, c),
 *              (d, e, f); mode 7: interactions [a, b) of user_list;
 *   item side: mode 0 .. 3 = that many interactions inline as (user row, slot) = (a, b), (c, d), (e, f);
 *              mode 7: interactions [a, b) of item_list.
 * `position` indexes the minibatch slices of envs / sample_weights; `slot` = rec_slot[position] = the interaction's
 * index in item_list order indexes the record array.  Every row of every
 * table appears exactly once: in a job or in the stream list (rows the minibatch does not touch: they get the
 * dense-Adam step with a zero gradient, rows_per_stream_task rows per workgroup, in either launch). */
typedef struct InvPrefRowPlan {
    int32_t n, lanes_per_group;               /* n = interactions of the minibatch (shard) */
    int32_t n_user_rounds, n_item_rounds, user_rounds_per_task, item_rounds_per_task;
    const int32_t *user_desc;                 /* [n_user_rounds][NG][8] */
    const int32_t *item_desc;                 /* [n_item_rounds][NG][8] */
    const int32_t *user_round_iters;          /* [n_user_rounds] interactions of the round's longest slice */
    const int32_t *user_list;                 /* [n][4] {item row, position, label bits, slot}, sorted by user row */
    const int32_t *item_list;                 /* [n][2] {user row, slot}, sorted by item row (entry s has slot s) */
    int32_t n_stream, rows_per_stream_task;
    const int32_t *stream_rows;               /* [n_stream] row id; bit 30 set: a row of the item tables */
    /* XCD-affine task order (speed only; any order gives the same results).  n_classes = 8 (the XCDs of an MI355X), or
     * 1 / 0 for the plain order.  Table rows are dealt to the classes in blocks of 64 rows, class(row) =
     * (row >> 6) % n_classes; rounds and streamed rows are grouped by class, and workgroup b of a launch runs tasks of
     * class b % n_classes only -- under the round-robin placement of workgroups over the XCDs that is the same XCD step
     * after step, so a row's parameters and Adam moments are still in that XCD's L2 when the next step reads them.
     * cls[c] = { first user round, user rounds (both multiples of user_rounds_per_task), first streamed row of launch 1
     * (index into stream_rows), streamed rows of launch 1, first item round, item rounds, first streamed row of
     * launch 2, streamed rows of launch 2 } of class c. */
    int32_t n_classes, rows_per_stream_task2;  /* streamed rows per workgroup in launch 2 (0: rows_per_stream_task) */
    int32_t cls[8][8];
    /* optional (NULL: "pull" form, launch 2 gathers partner rows + records): the "push" form.  push_slot[position] = the
     * interaction's index in item_list order.  Launch 1 then stores, per interaction, its two contribution rows to its
     * item's gradient at that slot of the workspace, and launch 2's item jobs sum contiguous rows: every item descriptor
     * must be of mode 7 (or 0) with [a, b) = the slice's slots.  One extra row write + read per interaction and table: for
     * minibatches whose rows are a small share of the step's bytes (plan.py decides). */
    const int32_t *push_slot;                 /* [n] */
    /* required: rec_slot[position] = the interaction's index in item_list order (its "slot").  Launch 1 stores the record
     * launch 2 consumes (pull form) at THAT index of the workspace, so that an item job's slice of records is contiguous
     * and read front to back; user_list's word 3 and the second word of every item-side id pair (item_list, inline
     * descriptors) hold the slot as well.  In push form push_slot is this same array. */
    const int32_t *rec_slot;                  /* [n] */
} InvPrefRowPlan;

/* Scratch of one planned step: the records + the partial slabs.  It needs no initialisation (every word is stored
 * before it is loaded) and carries nothing from one call to the next. */
size_t invpref_rows_workspace_bytes(const InvPrefTables *tables, const InvPrefRowPlan *plan);
/* lanes_per_group the kernels use for tables of this factor_num (16 / 16 / 32); a plan must be built for it.  Rows of
 * more than 128 floats (32 lanes) take the pull form only: push_slot must be NULL. */
int invpref_rows_lanes_per_group(const InvPrefTables *tables);

/* same contract as invpref_mstep_grad_hip, except that EVERY row of every table of `grads` is
 * OVERWRITTEN (rows the minibatch does not touch get zeros): no zeroing pass is needed.  There are no float
 * atomics on this path: every gradient and loss term is bitwise reproducible run to run (register accumulation,
 * fixed-order sums).  `scores` is ignored (the labels travel inside the plan) and may be NULL. */
int invpref_mstep_rows_grad_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                const int64_t *envs, const float *scores, const float *sample_weights,
                                int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                void *workspace, size_t workspace_bytes, void *stream);

/* M-step + optimizer.step() in one pass (train.py:94-157 entire): reads `tables`, writes the updated
 * parameters into `new_tables` (a second buffer: other workgroups still gather the old rows) and
 * updates exp_avg / exp_avg_sq in place; the gradient is never stored.  Same Adam rule as
 * invpref_adam_hip.  Single-GPU path (a row-sharded run must all-reduce the gradient first). */
int invpref_mstep_rows_adam_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                uint32_t flags, float *losses6, int64_t step, double lr, double beta1, double beta2,
                                double eps, void *workspace, size_t workspace_bytes, void *stream);

/* Profiling aid (bench.py's roofline figure): the same call, recording `mid_event` (a hipEvent_t) on the stream
 * BETWEEN the two launches, so each launch can be timed alone with events.  No global state. */
int invpref_mstep_rows_adam_profiled_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                         const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                         const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                         const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                         uint32_t flags, float *losses6, int64_t step, double lr, double beta1,
                                         double beta2, double eps, void *workspace, size_t workspace_bytes, void *stream,
                                         void *mid_event);

/* ---- the same pass for HIP-graph replay.  A captured launch freezes its kernel arguments, so the
 * per-step scalars cannot be passed by value: they live in one of two device slots, picked by the
 * frozen argument `slot` = (step & 1).  A pass for step s reads slot s&1 and fills the other slot with
 * step s+1 and table[s+1 - base] for its successor (no ticket, no ordering between workgroups: nobody
 * reads that slot before the next launch).
 *   table: device float[n][8], one row per step: the six Adam scalars (invpref_adam_schedule_fill() computes them on
 *          the host), then the gradient-reversal alpha of that step -- NaN, as the fill leaves it, means "the alpha of
 *          the call's coefficient block"; a caller running the alpha schedule of train.py:214-217 writes it here --
 *          then one unused float.
 *   state: device int32[32], slot p at state + 16p = {step (1-based), base = step that table[0] belongs to,
 *          table[step - base] (8 floats), unused...}.  Before the first pass (and after a refill / rebase) the caller
 *          writes slot (step & 1) for the step about to run.
 *   slot : parity of the step this call performs; consecutive calls alternate.
 * The caller keeps step - base inside [0, n). */
typedef struct InvPrefAdamSchedule {
    int32_t *state;
    const float *table;
    int32_t n;
    int32_t slot;
} InvPrefAdamSchedule;
int invpref_adam_schedule_fill(float *host_table, int64_t first_step, int64_t n, double lr, double beta1, double beta2,
                               double eps);
int invpref_mstep_rows_adam_sched_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                      const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                      const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                      const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                      uint32_t flags, float *losses6, const InvPrefAdamSchedule *sched,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- ONE launch per optimiser step: the evaluating side alternates (round 5).
 * The step's arithmetic is symmetric in users and items (models.py:307-326) and torch.optim.Adam (train.py:41, :155-157)
 * updates every row exactly once per step, so launch c of a run can be
 *     side S = users (c even) or items (c odd); T = the other side
 *     (i)   S's rows apply step c-1's update: the contribution rows T's jobs pushed for them in launch c-1 (summed in slot
 *           order, + the regulariser term), Adam with step c-1's scalars -- in registers;
 *     (ii)  S's jobs evaluate step c's interactions from their side: own rows now at post-(c-1), partner rows of T
 *           untouched by this launch and at post-(c-1) since T finished its own step c-1 rows in launch c-1;
 *     (iii) they apply their own step-c gradient on the spot (second Adam, same registers), store p, m, v ONCE, and push
 *           the interaction's two contribution rows for T (consumed by launch c+1).
 * The one grid-wide dependency left is the 2.5 KB of small tables (embed_env, classifier): fold workgroups, first in the
 * grid, sum launch c-1's partial slabs, apply Adam IN PLACE with write-through stores and publish a flag per fold block
 * (value = the step number); job workgroups do (i) meanwhile and read the tables with cache-bypassing loads once the
 * flags match.  The six loss terms of step c-1 are produced by that fold (losses6_prev).
 * A run starts from a state in which every row is up to date (first launch: plan->has_prev = 0) and ends with a FLUSH
 * launch (plan->has_cur = 0: T's rows apply the last step's update, the last fold).  Only between those two the
 * parameter tables are in an intermediate state; the tables are updated in place (no second parameter buffer).
 * Compiled for rows of up to 64 floats and up to four environments (the two-launch form covers the rest).
 *
 * An alt plan describes ONE launch: the current minibatch from side S (desc / list / push_slot as user_desc / user_list /
 * push_slot of InvPrefRowPlan, with S in the role of the users) and the PREVIOUS launch's pushes for S:
 *   pend[round][slot] = {a, b, count, 0}: this slice's share [a, b) of the row's pending contribution rows (slots of
 *           the previous minibatch in S-sorted order) and the row's interaction count in the previous minibatch;
 *   meta bit 31 of every slot of a round: some row of the round has pending rows (the slices then meet once more);
 *   stream[i] = {row, a, b, count}: rows of S without a job (no interaction now, few or no pending rows). */
typedef struct InvPrefAltPlan {
    int32_t side;                 /* 0: the user tables evaluate / are updated; 1: the item tables */
    int32_t has_prev, has_cur;    /* 0 / 1, see above */
    int32_t n, n_prev;            /* interactions of the current / previous minibatch */
    int32_t lanes_per_group;      /* 16 */
    int32_t slots_per_round;      /* NG: 16 (workgroups of 256 threads) or 32 (512 threads: up to 32 slices per row; a slice
                                   * count of 32 is stored as 0 in the descriptor's 5-bit field) */
    int32_t n_rounds, rounds_per_task;
    const int32_t *desc;          /* [n_rounds][NG][8] */
    const int32_t *pend;          /* [n_rounds][NG][4] */
    const int32_t *list;          /* [n][4] {partner row, position, label bits, 0} sorted by own row */
    const int32_t *push_slot;     /* [n] position -> slot in PARTNER-sorted order (where this launch pushes) */
    int32_t n_stream, rows_per_stream_task;
    const int32_t *stream;        /* [n_stream][4] */
    int32_t n_classes;
    int32_t cls[8][4];            /* per class: first round, rounds, first stream row, stream rows */
    int32_t n_partials_prev;      /* job tasks (= partial slabs) of the previous launch */
} InvPrefAltPlan;
/* workspace of a run: two halves (parity) of {contribution rows for n_cap interactions, partials_cap partial slabs} +
 * the fold flags.  Needs no initialisation. */
size_t invpref_alt_workspace_bytes(const InvPrefTables *tables, int32_t n_cap, int32_t partials_cap);
/* 1 if the shape runs the alternating form (factor_num <= 64, env_num <= 4) */
int invpref_alt_supported(const InvPrefTables *tables);
/* One launch.  tables / exp_avg / exp_avg_sq are updated IN PLACE.  envs / sample_weights: the CURRENT minibatch's
 * slices.  batch_norm / batch_norm_prev: mean() denominators of the current / previous minibatch.  losses6_prev: where
 * the previous step's six loss terms are ADDED (may be NULL when has_prev == 0).  Per-step scalars: from `sched` (graph
 * replay; slot = parity of the current step -- of the LAST step for a flush launch; state words 10 / 11 of a slot carry
 * the previous step's step_size / bc2_sqrt, maintained by these launches and written by the caller for the first slot)
 * or, sched == NULL, from (step, lr, betas, eps): `step` is the current step (the last one for a flush).
 * parity: which workspace half this launch writes (the other one holds the previous launch's output); consecutive
 * launches alternate.  Same flags / coefficient conventions as invpref_mstep_rows_adam_hip. */
int invpref_mstep_alt_hip(const InvPrefTables *tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                          const InvPrefAltPlan *plan, const int64_t *envs, const float *sample_weights,
                          int64_t batch_norm, int64_t batch_norm_prev, const InvPrefCoefs *coefs, uint32_t flags,
                          float *losses6_prev, int64_t step, double lr, double beta1, double beta2, double eps,
                          const InvPrefAdamSchedule *sched, void *workspace, size_t workspace_bytes, int32_t n_cap,
                          int32_t partials_cap, int32_t parity, void *stream);
/* device word the job workgroups set when a wait for the fold flags ran out (never in a healthy run): the int32 at this
 * byte offset of the workspace */
size_t invpref_alt_error_offset(const InvPrefTables *tables, int32_t n_cap, int32_t partials_cap);

/* The gradient-pass + stand-alone-Adam sequence (multi-GPU: an all-reduce sits between the two; single GPU: rows of
 * more than 128 floats) for HIP-graph replay.  invpref_mstep_rows_grad_sched_hip reads the step's slot (a scheduled
 * gradient-reversal alpha, train.py:214-217) and leaves the schedule where it is; invpref_adam_ranges_sched_hip --
 * the step's last launch -- takes its Adam scalars from the slot and fills the other slot for the next step.  Same
 * contracts otherwise as invpref_mstep_rows_grad_hip / invpref_adam_ranges_hip below. */
int invpref_mstep_rows_grad_sched_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                      const int64_t *envs, const float *scores, const float *sample_weights,
                                      int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                      const InvPrefAdamSchedule *sched, void *workspace, size_t workspace_bytes,
                                      void *stream);
int invpref_adam_ranges_sched_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *offsets,
                                  const int64_t *lengths, int32_t n_ranges, const InvPrefAdamSchedule *sched,
                                  int zero_grad, void *stream);

/* ---- backward of forward(): replaces autograd through InvPref*.forward + ReverseLayerF
 * (models.py:307-326 / :448-467, functions.py:7-16) for callers that build their own loss on the
 * unfused outputs (e.g. the reference's untouched train.py).  Upstream gradients d_* have the shapes
 * of the forward outputs; any of them may be NULL (zeros).  ADDS into `grads`. */
int invpref_backward_hip(const InvPrefTables *tables, const InvPrefTables *grads, const int64_t *users,
                         const int64_t *items, const int64_t *envs, int64_t B, uint32_t flags, float alpha,
                         const float *d_invariant_score, const float *d_env_aware_score, const float *d_env_outputs,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- predict: replaces InvPrefImplicit.predict (models.py:393-407): out[n_users, item_num] =
 * sigmoid(user_table[users] . item_table^T) (apply_sigmoid=0: raw dot products). */
int invpref_predict_hip(const float *user_table, const float *item_table, const int64_t *users, int64_t n_users,
                        int64_t item_num, int64_t factor_num, int apply_sigmoid, float *out, void *stream);

/* ---- evaluation (the first "next" row): the per-user part of
 * ImplicitTestManager.evaluate_batch (evaluate.py:88-120) on a rating matrix produced by
 * invpref_predict_hip.  CSR inputs (int32, device): for test user j (row j of `ratings`)
 *   mask_items[mask_ptr[j] .. mask_ptr[j+1])       train items, rating := -1024      (evaluate.py:94-101)
 *   highlight_items[...] (highlight_ptr may be NULL)  item pool, rating += 1024      (evaluate.py:103-111)
 *   truth_items[truth_ptr[j] .. truth_ptr[j+1])    SORTED ground-truth items         (evaluate.py:11-19)
 * Outputs [n_users, k]: the top-k item ids in descending rating order (lowest id first among equal
 * ratings) and 1.0/0.0 hit labels (get_label).  ratings is not modified.  k <= 64, k <= n_items <= 400000
 * (up to 10 240 items four rating rows are staged in LDS per workgroup; beyond that the row stays in global
 * memory and LDS holds bit sets over the items -- MIND's 51 283 items included). */
int invpref_eval_topk_hip(const float *ratings, int64_t n_users, int64_t n_items, const int32_t *mask_ptr,
                          const int32_t *mask_items, const int32_t *highlight_ptr, const int32_t *highlight_items,
                          const int32_t *truth_ptr, const int32_t *truth_items, int32_t k, int32_t *out_items,
                          float *out_hits, void *stream);

/* ---- ExplicitTestManager.evaluate (evaluate.py:187-212): out2 (device double[2]) = {sum (pred-target)^2,
 * sum |pred-target|}; mse / rmse / mae follow on the host. */
int invpref_eval_error_sums_hip(const float *pred, const float *target, int64_t n, double *out2, void *stream);

/* ---- ImplicitTrainStaticPopularityManager.static_pop (train.py:509-571): for every environment e, ten float64
 * means over the interactions currently assigned to e (their `users`/`items`), in the reference's key order:
 *   0 users_cnt_weight      mean over interactions of user_cnt[u]           1 items_cnt_weight      ... item_cnt[i]
 *   2 users_normalize_cnt_weight   ... user_cnt_norm[u]                      3 items_normalize_cnt_weight ... item_cnt_norm[i]
 *   4 users_cnt             mean over the DISTINCT users of e of user_cnt   5 items_cnt             (distinct items)
 *   6 users_normalize_cnt   distinct users, user_cnt_norm                   7 items_normalize_cnt   (distinct items)
 *   8 pair_cnt_add          mean of user_cnt[u] + item_cnt[i]               9 pair_normalize_cnt_multiply  mean of norm[u]*norm[i]
 * user_cnt / item_cnt / *_norm are the loader's tables (dataloader.py:273-291).  An empty environment gives NaN
 * (np.mean of an empty array).  out: device double[env_num][10].  Integer sums are exact. */
size_t invpref_static_pop_workspace_bytes(int64_t user_num, int64_t item_num, int64_t env_num);
int invpref_static_pop_hip(const int64_t *users, const int64_t *items, const int64_t *envs, int64_t n, int64_t user_num,
                           int64_t item_num, int64_t env_num, const int64_t *user_cnt, const int64_t *item_cnt,
                           const double *user_cnt_norm, const double *item_cnt_norm, double *out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- packed exchange of a row-sharded optimiser step (the reference is single-process: its
 * loss.backward() / optimizer.step(), train.py:155-157, see the whole minibatch).  Only the rows of the four big tables
 * that the GLOBAL minibatch touches carry a gradient; every other row is zero on every rank.  pack: copies those rows
 * (row_offsets[r] = first float of row r inside `flat`, D floats each) and then flat[tail_offset, tail_offset + tail_len)
 * (the small tables) into `packed` ((n_rows * D + tail_len) floats), which the caller all-reduces; unpack: the reverse copy.
 * vec_ok != 0: every row offset is a multiple of 4 floats (float4 copies when D % 4 == 0 and the buffers are 16-byte
 * aligned as well). */
int invpref_pack_rows_hip(const float *flat, const int64_t *row_offsets, int64_t n_rows, int32_t D, int64_t tail_offset,
                          int64_t tail_len, float *packed, int vec_ok, void *stream);
int invpref_unpack_rows_hip(float *flat, const int64_t *row_offsets, int64_t n_rows, int32_t D, int64_t tail_offset,
                            int64_t tail_len, const float *packed, int vec_ok, void *stream);

/* ---- dense Adam: replaces optimizer.zero_grad() + optimizer.step() of torch.optim.Adam with
 * default betas/eps (train.py:41, :155-157) over one flat fp32 buffer of n parameters.
 * step is 1-based.  zero_grad != 0 also clears grad (the next step's zero_grad()).  The buffers need only be
 * float-aligned: the float4 body starts at their first common 16-byte boundary (buffers misaligned differently
 * from one another are processed one float at a time). */
int invpref_adam_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                     double lr, double beta1, double beta2, double eps, int zero_grad, void *stream);
/* The same step over 1..4 pieces [offsets[r], offsets[r] + lengths[r]) of the flat buffers in one launch (host
 * arrays; every offset and length a multiple of 4 floats).  A user-sharded rank updates its own rows of the two
 * user tables and the shared tail (item tables, embed_env, classifier) this way. */
int invpref_adam_ranges_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *offsets,
                            const int64_t *lengths, int32_t n_ranges, int64_t step, double lr, double beta1, double beta2,
                            double eps, int zero_grad, void *stream);

/* ---- E-step: replaces cluster() / cluster_a_batch() / cluster_predict() (train.py:235-259,
 * :169-202; models.py:409-411) over all N interactions, followed by stat_envs() (train.py:268-280).
 * eps_rows: optional [N, env_num] tie-break rows already gathered per sample (train.py:192-196), or NULL.
 * old_envs may alias new_envs.  Outputs: new_envs[N]; counts[env_num] (int64); diff[1] (int64, number of
 * changed assignments, train.py:256-257); class_weights[env_num] and sample_weights[N] (train.py:274-278),
 * either may be NULL.  workspace as above. */
size_t invpref_estep_workspace_bytes(const InvPrefTables *tables, int64_t N);
int invpref_estep_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items,
                      const float *scores, int64_t N, uint32_t flags, const float *eps_rows,
                      const int64_t *old_envs, int64_t *new_envs, int64_t *counts, int64_t *diff,
                      float *class_weights, float *sample_weights, void *workspace, size_t workspace_bytes,
                      void *stream);

/* The same E-step with the reference's DEFAULT tie-break (cluster_use_random_sort=True: train.py:86-92 builds all E!
 * permutations of eps_base = [1e-10, 1e-11, ...] in itertools.permutations order, train.py:192-196 adds row
 * np.random.randint(0, E!) to every interaction's distances) WITHOUT the E! x E table and without an N x E table of
 * gathered rows: perm_index[i] (device; index_bytes = 1, 4 or 8 bytes per entry, unsigned / int32 / int64 -- wide enough
 * for E! - 1) is the permutation row drawn for interaction i and eps_base (HOST, float[env_num]) the vector that is
 * permuted; the row is unranked on the device.  Bit for bit the result of invpref_estep_hip on the gathered rows.
 * perm_index may also be PINNED HOST memory the device can address (hipHostMalloc): up to seven environments the kernel
 * reads every workgroup's indices in one burst, so no copy is needed in front of it. */
int invpref_estep_perm_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                           int64_t N, uint32_t flags, const void *perm_index, int index_bytes, const float *eps_base,
                           const int64_t *old_envs, int64_t *new_envs, int64_t *counts, int64_t *diff,
                           float *class_weights, float *sample_weights, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ---- cluster() + the stat_envs() that always follows it (train.py:235-259, :268-280, called together at train.py:329-330) as
 * ONE launch (round 6): the assignment kernel's workgroups publish their count slabs and take a ticket, the
 * last one folds them into counts[env_num], diff[1] (cluster()'s diff_num) and class_weights[env_num] =
 * min(count + 1, N - 1) / N.  No N-length sample_weights array is produced: the planned M-step entry points form
 * sample_weights[i] = class_weights[envs[i]] themselves under INVPREF_WEIGHTS_BY_ENV.
 *   envs        in place: the old assignment of row i is read, the new one written, by the same lane;
 *   perm_index  NULL: plain argmin (cluster_use_random_sort=False); else as invpref_estep_perm_hip, with
 *   perm_table  (device, optional, env_num <= 7) the E! packed permutation rows made once by invpref_perm_table_fill -- the
 *               workgroups then load the table instead of unranking it;
 *   state       device int32[INVPREF_ESTEP_STATE_INTS] that must be ZERO before the first call and is left as the next call
 *               needs it ([0] top ticket, [1] E-steps so far = ring position, [32 + 32 s] the ticket of shard s: 2 048
 *               workgroups finishing together would queue ~25 us on ONE word; [33 + 32 s .. 33 + 32 s + env_num] the shard's
 *               counts and diff, added by its workgroups with integer atomics and read -- then zeroed -- by the launch's last one);
 *   ring        optional int64[ring_cap][env_num + 1]: the call writes {counts, diff} to row (calls so far) % ring_cap and counts
 *               the call in state[1] -- a captured launch's arguments are frozen, so a replayed E-step keeps its results apart
 *               this way without a copy behind every replay;
 *   counts / diff / class_weights  optional direct outputs (NULL: skipped).
 * Single rank only (the class weights need the GLOBAL counts: a sharded rank uses invpref_estep_perm_hip / invpref_estep_hip,
 * all-reduces the counts and calls invpref_sample_weights_hip). */
#define INVPREF_ESTEP_STATE_INTS (32 + 32 * 32)
int invpref_estep_fused_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                            int64_t N, uint32_t flags, const void *perm_index, int index_bytes, const float *eps_base,
                            const uint32_t *perm_table, int64_t *envs, int32_t *state, int64_t *ring, int32_t ring_cap,
                            int64_t *counts, int64_t *diff, float *class_weights, void *workspace, size_t workspace_bytes,
                            void *stream);
/* host helper: host_table[r] = permutation row r of train.py:86-92 (itertools.permutations order), 4 bits per position;
 * env_num <= 7.  Returns the number of rows (env_num!) or INVPREF_EINVAL. */
int invpref_perm_table_fill(int32_t env_num, uint32_t *host_table);

/* ---- stat_envs alone (train.py:268-280), e.g. before the first epoch (train.py:297). */
int invpref_stat_envs_hip(const int64_t *envs, int64_t N, int64_t env_num, int64_t *counts, float *class_weights,
                          float *sample_weights, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the weight half of stat_envs (train.py:274-278) from counts that are already global
 * (row-sharded runs all-reduce the per-rank counts first): class_weights[e] = min(counts[e]+1,
 * N_total-1)/N_total, sample_weights[i] = class_weights[envs[i]] for the N_local local rows. */
int invpref_sample_weights_hip(const int64_t *envs, int64_t N_local, const int64_t *counts, int64_t N_total,
                               int64_t env_num, float *class_weights, float *sample_weights, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* INVPREF_HIP_H */
