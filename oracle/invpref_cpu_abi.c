/* TEST INFRASTRUCTURE -- not part of the product, never loaded by it.
 *
 * CPU twins of the core-path entry points of include/invpref_hip.h (SURVEY section 8(b) level 4, section 7 step 2:
 * "the same C ABI, two back ends"): identical argument lists -- the InvPrefTables / InvPrefCoefs structs, flags, the
 * trailing workspace and stream arguments (ignored here) -- over HOST memory, implemented with the oracle's f32 routines
 * (invpref_oracle.c).  tests/test_cpu_abi_twins.py drives both libraries through ONE set of ctypes prototypes: the twins
 * against the golden vectors on the CPU, and the HIP library against the twins on the GPU.
 * The product has no CPU path: libinvpref_hip.so neither links nor loads this file. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/invpref_hip.h"

typedef struct {
    int64_t U, I, E, D;
    const float *Pu, *Qi, *Pa, *Qa, *Ev, *W, *b;
} oracle_tables;
typedef struct {
    float *Pu, *Qi, *Pa, *Qa, *Ev, *W, *b;
} oracle_grads;
void oracle_forward_f32(const oracle_tables *t, const int64_t *u, const int64_t *v, const int64_t *e, int64_t B,
                        uint32_t flags, float *inv, float *envaware, float *envout);
void oracle_mstep_f32(const oracle_tables *t, const oracle_grads *g, const int64_t *u, const int64_t *v, const int64_t *e,
                      const float *y, const float *w, int64_t B, int64_t Bnorm, const double *coefs, uint32_t flags,
                      int include_dense_reg, double *losses);
void oracle_adam_f32(float *p, const float *g, float *m, float *vv, int64_t n, int64_t step, double lr, double beta1,
                     double beta2, double eps);
void oracle_estep_f32(const oracle_tables *t, const int64_t *u, const int64_t *v, const float *y, int64_t N, uint32_t flags,
                      const float *eps_rows, const int64_t *old_envs, int64_t *new_envs, int64_t *counts, int64_t *diff,
                      float *dist_out);
void oracle_stat_envs_f32(const int64_t *envs, int64_t N, int64_t E, int64_t *counts, float *class_w, float *sample_w);

static int tables_ok(const InvPrefTables *t) {
    if (!t || t->user_num < 0 || t->item_num < 0 || t->env_num <= 0 || t->factor_num <= 0) return INVPREF_EINVAL;
    if (t->factor_num > INVPREF_MAX_FACTORS || t->env_num > INVPREF_MAX_ENVS) return INVPREF_EUNSUPPORTED;
    if (!t->embed_user_invariant || !t->embed_item_invariant || !t->embed_user_env_aware || !t->embed_item_env_aware ||
        !t->embed_env || !t->classifier_weight || !t->classifier_bias)
        return INVPREF_EINVAL;
    return 0;
}
static oracle_tables as_oracle(const InvPrefTables *t) {
    oracle_tables o = {t->user_num, t->item_num, t->env_num, t->factor_num, t->embed_user_invariant,
                       t->embed_item_invariant, t->embed_user_env_aware, t->embed_item_env_aware, t->embed_env,
                       t->classifier_weight, t->classifier_bias};
    return o;
}

int invpref_forward_cpu(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const int64_t *envs,
                        int64_t B, uint32_t flags, float *invariant_score, float *env_aware_score, float *env_outputs,
                        void *stream) {
    (void)stream;
    int rc = tables_ok(tables);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!users || !items || !envs || !invariant_score || !env_aware_score || !env_outputs)))
        return INVPREF_EINVAL;
    const oracle_tables t = as_oracle(tables);
    oracle_forward_f32(&t, users, items, envs, B, flags, invariant_score, env_aware_score, env_outputs);
    return 0;
}

size_t invpref_mstep_workspace_bytes_cpu(const InvPrefTables *tables, int64_t B) { (void)tables; (void)B; return 0; }

int invpref_mstep_grad_cpu(const InvPrefTables *tables, const InvPrefTables *grads, const int64_t *users,
                           const int64_t *items, const int64_t *envs, const float *scores, const float *sample_weights,
                           int64_t B, int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                           void *workspace, size_t workspace_bytes, void *stream) {
    (void)workspace; (void)workspace_bytes; (void)stream;
    int rc = tables_ok(tables);
    if (rc) return rc;
    const int no_grad = (flags & INVPREF_NO_GRAD) != 0;
    if (!coefs || !losses6 || B < 0 || batch_norm <= 0 || (B > 0 && (!users || !items || !envs || !scores)) ||
        (!no_grad && !grads))
        return INVPREF_EINVAL;
    if ((flags & (INVPREF_REWEIGHT_REC | INVPREF_REWEIGHT_CLS)) && B > 0 && !sample_weights) return INVPREF_EINVAL;
    const oracle_tables t = as_oracle(tables);
    const int64_t U = t.U, I = t.I, E = t.E, D = t.D;
    /* gradients are ADDED (as the HIP entry point does): the oracle adds into what it is handed */
    float *scratch = NULL;
    oracle_grads g;
    if (no_grad) {
        scratch = (float *)calloc((size_t)(2 * (U + I) * D + 2 * E * D + E), sizeof(float));
        if (!scratch) return INVPREF_EWORKSPACE;
        float *p = scratch;
        g.Pu = p; p += U * D; g.Qi = p; p += I * D; g.Pa = p; p += U * D; g.Qa = p; p += I * D;
        g.Ev = p; p += E * D; g.W = p; p += E * D; g.b = p;
    } else {
        g.Pu = grads->embed_user_invariant; g.Qi = grads->embed_item_invariant; g.Pa = grads->embed_user_env_aware;
        g.Qa = grads->embed_item_env_aware; g.Ev = grads->embed_env; g.W = grads->classifier_weight;
        g.b = grads->classifier_bias;
    }
    float *ones = NULL;
    if (!sample_weights && B > 0) {
        ones = (float *)malloc(sizeof(float) * (size_t)B);
        if (!ones) { free(scratch); return INVPREF_EWORKSPACE; }
        for (int64_t i = 0; i < B; i++) ones[i] = 1.0f;
    }
    const double c6[6] = {coefs->invariant_coe, coefs->env_aware_coe, coefs->env_coe, coefs->L2_coe, coefs->L1_coe,
                          coefs->alpha};
    double l6[6] = {0, 0, 0, 0, 0, 0};
    oracle_mstep_f32(&t, &g, users, items, envs, scores, sample_weights ? sample_weights : ones, B, batch_norm, c6,
                     flags & 31u, (flags & INVPREF_DENSE_REG) != 0, l6);
    for (int k = 0; k < 6; k++) losses6[k] += (float)l6[k];   /* ADDED to, as in the header */
    free(ones);
    free(scratch);
    return 0;
}

int invpref_adam_cpu(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, double lr,
                     double beta1, double beta2, double eps, int zero_grad, void *stream) {
    (void)stream;
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return INVPREF_EINVAL;
    oracle_adam_f32(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps);
    if (zero_grad && n > 0) memset(grad, 0, sizeof(float) * (size_t)n);
    return 0;
}

size_t invpref_estep_workspace_bytes_cpu(const InvPrefTables *tables, int64_t N) { (void)tables; (void)N; return 0; }

int invpref_estep_cpu(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                      int64_t N, uint32_t flags, const float *eps_rows, const int64_t *old_envs, int64_t *new_envs,
                      int64_t *counts, int64_t *diff, float *class_weights, float *sample_weights, void *workspace,
                      size_t workspace_bytes, void *stream) {
    (void)workspace; (void)workspace_bytes; (void)stream;
    int rc = tables_ok(tables);
    if (rc) return rc;
    if (N < 0 || !counts || !diff || (N > 0 && (!users || !items || !scores || !old_envs || !new_envs))) return INVPREF_EINVAL;
    const oracle_tables t = as_oracle(tables);
    /* old_envs may alias new_envs: the oracle reads old_envs[i] before it writes new_envs[i] */
    oracle_estep_f32(&t, users, items, scores, N, flags & 1u, eps_rows, old_envs, new_envs, counts, diff, NULL);
    if (class_weights) {
        int64_t tmp[INVPREF_MAX_ENVS];
        oracle_stat_envs_f32(new_envs, N, t.E, tmp, class_weights, sample_weights);
    }
    return 0;
}

int invpref_stat_envs_cpu(const int64_t *envs, int64_t N, int64_t env_num, int64_t *counts, float *class_weights,
                          float *sample_weights, void *workspace, size_t workspace_bytes, void *stream) {
    (void)workspace; (void)workspace_bytes; (void)stream;
    if (N < 0 || env_num <= 0 || env_num > INVPREF_MAX_ENVS || !counts || !class_weights || (N > 0 && !envs))
        return INVPREF_EINVAL;
    oracle_stat_envs_f32(envs, N, env_num, counts, class_weights, sample_weights);
    return 0;
}
