/* invpref_ingest.h -- C ABI of the host-side data ingest.
 *
 * Replaces the CSV reading and per-line python parsing of the reference's data loaders
 * (dataloader.py:124-128, :395-399 `pd.read_csv(...).values`; utils.py:208-234
 * `analyse_interaction_from_text`; utils.py:236-251 `analyse_user_interacted_set`) with one native pass:
 * numeric CSV -> dense array, (user, item) pairs -> CSR of the distinct items of every user.
 * Plain C: pointers and sizes only.  Host code, no GPU involved.  Returns 0 or a negative INVPREF_INGEST_E* code. */
#ifndef INVPREF_INGEST_H
#define INVPREF_INGEST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define INVPREF_INGEST_EIO (-1)     /* cannot open / map the file */
#define INVPREF_INGEST_EPARSE (-2)  /* a field is not a number, or a line has the wrong number of fields */
#define INVPREF_INGEST_EINVAL (-3)

int invpref_ingest_abi_version(void);

/* First pass: number of data rows (blank lines skipped), number of comma-separated columns, and whether
 * the first line is a header (its first field is not a number), as pd.read_csv infers it. */
int invpref_csv_shape(const char *path, int64_t *rows, int32_t *cols, int32_t *has_header);

/* Second pass: every field as a double, row-major [rows, cols]; `rows` / `cols` / `has_header` as
 * reported by invpref_csv_shape.  n_threads <= 0: one per hardware thread (at most 16). */
int invpref_csv_read_f64(const char *path, int64_t rows, int32_t cols, int32_t has_header, double *out,
                         int32_t n_threads);

/* analyse_user_interacted_set (utils.py:236-251) as CSR: for user u the sorted DISTINCT items of its
 * pairs are indices[indptr[u] .. indptr[u+1]).  indptr has n_users + 1 entries, indices room for n.
 * Users must lie in [0, n_users).  Returns the number of indices written (>= 0) or a negative code. */
int64_t invpref_csr_sets(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t *indptr,
                         int64_t *indices);

#ifdef __cplusplus
}
#endif
#endif
