// Host-side data ingest (include/invpref_ingest.h): numeric CSV -> dense doubles, pairs -> CSR sets.
// mmap + one thread per byte range (ranges cut at line ends); no locale, no allocation per field.
#include "../../include/invpref_ingest.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char *)m;
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

inline bool blank_line(const char *b, const char *e) {
    for (; b < e; b++)
        if (*b != ' ' && *b != '\t' && *b != '\r') return false;
    return true;
}

// one field [b, e) -> double.  Plain decimal integers / fractions are converted directly (exactly, for
// up to 15 significant digits); anything else goes through strtod on a bounded copy.
inline bool parse_field(const char *b, const char *e, double *out) {
    while (b < e && (*b == ' ' || *b == '\t')) b++;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) e--;
    if (b == e) return false;
    const char *q = b;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; q++; }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool simple = q < e;
    for (; q < e && *q >= '0' && *q <= '9'; q++) { mant = mant * 10 + (uint64_t)(*q - '0'); digits++; }
    if (q < e && *q == '.') {
        q++;
        for (; q < e && *q >= '0' && *q <= '9'; q++) { mant = mant * 10 + (uint64_t)(*q - '0'); digits++; frac++; }
    }
    if (q != e || digits == 0 || digits > 15 || frac > 15) simple = false;
    if (simple) {
        static const double p10[16] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const double v = (double)mant / p10[frac];  // both exact below 2^53: the quotient is correctly rounded
        *out = neg ? -v : v;
        return true;
    }
    char buf[64];
    const size_t len = (size_t)(e - b);
    if (len >= sizeof(buf)) return false;
    memcpy(buf, b, len);
    buf[len] = 0;
    char *endp = nullptr;
    const double v = strtod(buf, &endp);
    if (endp != buf + len) return false;
    *out = v;
    return true;
}

inline int count_cols(const char *b, const char *e) {
    int c = 1;
    for (; b < e; b++) c += (*b == ',');
    return c;
}

struct Range { size_t lo, hi; int64_t rows; };

// [start, end of file) cut into at most n_threads byte ranges that end right after a newline
std::vector<Range> cut(const Mapped &m, size_t start, int n_threads) {
    std::vector<Range> r;
    const size_t total = m.n - start;
    size_t lo = start;
    for (int t = 0; t < n_threads && lo < m.n; t++) {
        size_t hi = (t == n_threads - 1) ? m.n : std::max(lo, start + total * (size_t)(t + 1) / (size_t)n_threads);
        if (hi < m.n) {
            const char *nl = (const char *)memchr(m.p + hi, '\n', m.n - hi);
            hi = nl ? (size_t)(nl - m.p) + 1 : m.n;
        }
        if (hi > lo) r.push_back({lo, hi, 0});
        lo = hi;
    }
    return r;
}

int threads_of(int n_threads, size_t bytes) {
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    const int by_size = (int)std::max<size_t>(1, bytes >> 16);  // at least 64 KiB per thread
    return std::max(1, std::min(n_threads, by_size));
}

template <typename F>
void for_lines(const Mapped &m, const Range &r, F &&f) {
    size_t b = r.lo;
    while (b < r.hi) {
        const char *nl = (const char *)memchr(m.p + b, '\n', r.hi - b);
        const size_t e = nl ? (size_t)(nl - m.p) : r.hi;
        if (!blank_line(m.p + b, m.p + e)) f(m.p + b, m.p + e);
        b = e + 1;
    }
}

size_t first_line_end(const Mapped &m) {
    const char *nl = (const char *)memchr(m.p, '\n', m.n);
    return nl ? (size_t)(nl - m.p) : m.n;
}

bool header_of(const Mapped &m) {
    if (m.n == 0) return false;
    const size_t e = first_line_end(m);
    const char *c = (const char *)memchr(m.p, ',', e);
    double v;
    return !parse_field(m.p, c ? c : m.p + e, &v);
}

}  // namespace

extern "C" {

int invpref_ingest_abi_version(void) { return 1; }

int invpref_csv_shape(const char *path, int64_t *rows, int32_t *cols, int32_t *has_header) {
    if (!path || !rows || !cols || !has_header) return INVPREF_INGEST_EINVAL;
    Mapped m;
    if (!m.open(path)) return INVPREF_INGEST_EIO;
    *rows = 0; *cols = 0; *has_header = 0;
    if (m.n == 0) return 0;
    const bool hdr = header_of(m);
    *has_header = hdr;
    const size_t fe = first_line_end(m);
    *cols = count_cols(m.p, m.p + fe);
    const size_t start = hdr ? std::min(m.n, fe + 1) : 0;
    auto ranges = cut(m, start, threads_of(0, m.n - start));
    std::vector<std::thread> th;
    for (auto &r : ranges)
        th.emplace_back([&m, &r] { for_lines(m, r, [&r](const char *, const char *) { r.rows++; }); });
    for (auto &t : th) t.join();
    for (auto &r : ranges) *rows += r.rows;
    return 0;
}

int invpref_csv_read_f64(const char *path, int64_t rows, int32_t cols, int32_t has_header, double *out,
                         int32_t n_threads) {
    if (!path || rows < 0 || cols <= 0 || (rows > 0 && !out)) return INVPREF_INGEST_EINVAL;
    Mapped m;
    if (!m.open(path)) return INVPREF_INGEST_EIO;
    const size_t start = has_header ? std::min(m.n, first_line_end(m) + 1) : 0;
    auto ranges = cut(m, start, threads_of(n_threads, m.n - start));
    // pass 1: rows per range -> output offsets; pass 2: parse
    {
        std::vector<std::thread> th;
        for (auto &r : ranges)
            th.emplace_back([&m, &r] { for_lines(m, r, [&r](const char *, const char *) { r.rows++; }); });
        for (auto &t : th) t.join();
    }
    int64_t total = 0;
    std::vector<int64_t> first(ranges.size());
    for (size_t i = 0; i < ranges.size(); i++) { first[i] = total; total += ranges[i].rows; }
    if (total != rows) return INVPREF_INGEST_EPARSE;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (size_t i = 0; i < ranges.size(); i++)
        th.emplace_back([&, i] {
            double *dst = out + first[i] * cols;
            for_lines(m, ranges[i], [&](const char *b, const char *e) {
                int c = 0;
                const char *f = b;
                while (true) {
                    const char *comma = (const char *)memchr(f, ',', (size_t)(e - f));
                    const char *fe = comma ? comma : e;
                    if (c >= cols || !parse_field(f, fe, dst + c)) { bad = 1; return; }
                    c++;
                    if (!comma) break;
                    f = comma + 1;
                }
                if (c != cols) bad = 1;
                dst += cols;
            });
        });
    for (auto &t : th) t.join();
    return bad ? INVPREF_INGEST_EPARSE : 0;
}

int64_t invpref_csr_sets(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t *indptr,
                         int64_t *indices) {
    if (n < 0 || n_users < 0 || !indptr || (n > 0 && (!users || !items || !indices))) return INVPREF_INGEST_EINVAL;
    std::vector<int64_t> cnt((size_t)n_users + 1, 0);
    for (int64_t i = 0; i < n; i++) {
        if (users[i] < 0 || users[i] >= n_users) return INVPREF_INGEST_EINVAL;
        cnt[(size_t)users[i] + 1]++;
    }
    for (int64_t u = 0; u < n_users; u++) cnt[(size_t)u + 1] += cnt[(size_t)u];
    std::vector<int64_t> tmp((size_t)n), fill(cnt.begin(), cnt.end() - 1);
    for (int64_t i = 0; i < n; i++) tmp[(size_t)fill[(size_t)users[i]]++] = items[i];
    int64_t w = 0;
    for (int64_t u = 0; u < n_users; u++) {
        int64_t *b = tmp.data() + cnt[(size_t)u], *e = tmp.data() + cnt[(size_t)u + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        indptr[u] = w;
        for (int64_t *q = b; q < e; q++) indices[w++] = *q;
    }
    indptr[n_users] = w;
    return w;
}

}  // extern "C"
