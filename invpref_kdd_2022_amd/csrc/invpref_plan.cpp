// Host-side row-plan builder (include/invpref_plan.h): the native twin of invpref_kdd_2022_amd/plan.py's numpy reference
// implementation (build_row_plan / _side_rounds), producing the same int32 arrays byte for byte from the same resolved
// parameters.  Two stable counting sorts, then one pass per XCD class and side; build_many runs minibatches on threads.
// (The reference's seam: utils.mini_batch, utils.py:12-19 -- the minibatches are the same contiguous slices every epoch.)
#include "../../include/invpref_plan.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct InvPrefHostPlan {
    std::vector<int32_t> arr[8];   // (alt plans use 0 .. 5)
    // the big arrays (lists, descriptors, push slots) live in UNINITIALISED storage: every element is written exactly once by
    // the builder, and zero-filling 0.5 GB first costs as much as the sort itself
    std::unique_ptr<int32_t[]> big[8];
    size_t big_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t *alloc(int which, size_t n) {
        big[which].reset(new int32_t[n ? n : 1]);
        big_n[which] = n;
        return big[which].get();
    }
};

namespace {

constexpr int kThreads = 256;
constexpr unsigned kMaxThreads = 32;   // host threads one build spreads over at most
constexpr int32_t kItemBit = 1 << 30;
constexpr int kModeList = 7;
constexpr int kClassShift = 6;

inline int64_t cdiv_i(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct Side {
    int64_t n_rows = 0;
    std::vector<int64_t> cnt, ptr;       // per row: interactions, first position in the sorted order
};

// run fn(0 .. count - 1) on up to `threads` threads
template <typename F>
void parallel_for(int count, int threads, F fn) {
    threads = std::max(1, std::min(threads, count));
    if (threads == 1) {
        for (int i = 0; i < count; i++) fn(i);
        return;
    }
    // (an exception on a worker thread -- std::bad_alloc of a histogram or of a 0.7 GB list -- would end the process in
    //  std::terminate; so would unwinding past joinable threads: every worker catches, every thread is joined, and the first
    //  failure is rethrown on the calling thread, where the C entry points turn it into a NULL handle)
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first;
    std::mutex mu;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count || failed.load()) return;
            try {
                fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                if (!first) first = std::current_exception();
                failed.store(true);
                return;
            }
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < threads; t++) th.emplace_back(work);
    } catch (...) {   // (no more threads to be had: the ones that started and this one do the work)
    }
    work();
    for (auto &t : th) t.join();
    if (first) std::rethrow_exception(first);
}

// Stable counting sort of the minibatch by row, on `threads` threads: emit(i, j) is called once per interaction i with its
// position j in the sorted order (equal rows keep their minibatch order).  Thread t counts and places the t-th contiguous
// chunk of the minibatch; its first position inside a row is the row's start + what the chunks before it hold of that row.
template <typename Emit>
bool sort_side(const int64_t *rows, int64_t n, int64_t n_rows, Side &s, int threads, Emit emit) {
    s.n_rows = n_rows;
    s.cnt.assign((size_t)n_rows, 0);
    s.ptr.assign((size_t)n_rows + 1, 0);
    // (one int32 histogram over the rows per thread: at most 256 MB of them)
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(threads, n >> 16), (int64_t(64) << 20) / std::max<int64_t>(n_rows, 1)));
    const int64_t chunk = T > 1 ? cdiv_i(n, T) : n;
    std::vector<std::vector<int32_t>> at((size_t)T);
    std::atomic<int> invalid{0};
    parallel_for(T, T, [&](int t) {
        std::vector<int32_t> &h = at[(size_t)t];
        h.assign((size_t)n_rows, 0);
        const int64_t a = std::min(n, t * chunk), b = std::min(n, a + chunk);
        bool bad = false;
        for (int64_t i = a; i < b; i++) {
            const uint64_t r = (uint64_t)rows[i];
            if (r < (uint64_t)n_rows) h[(size_t)r]++;
            else bad = true;
        }
        if (bad) invalid.store(1);
    });
    if (invalid.load()) return false;   // a row id out of range
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t c = 0;
        for (int t = 0; t < T; t++) c += at[(size_t)t][(size_t)r];
        s.cnt[(size_t)r] = c;
        s.ptr[(size_t)r + 1] = s.ptr[(size_t)r] + c;
    }
    parallel_for(T, T, [&](int blk) {   // counts -> first positions (rows split over the threads)
        const int64_t rc = cdiv_i(n_rows, T), ra = std::min(n_rows, blk * rc), rb = std::min(n_rows, ra + rc);
        for (int64_t r = ra; r < rb; r++) {
            int64_t pos = s.ptr[(size_t)r];
            for (int t = 0; t < T; t++) {
                const int32_t c = at[(size_t)t][(size_t)r];
                at[(size_t)t][(size_t)r] = (int32_t)pos;
                pos += c;
            }
        }
    });
    parallel_for(T, T, [&](int t) {
        std::vector<int32_t> &h = at[(size_t)t];
        const int64_t a = std::min(n, t * chunk), b = std::min(n, a + chunk);
        for (int64_t i = a; i < b; i++) emit(i, (int64_t)h[(size_t)rows[i]]++);
    });
    return true;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// plan.py: _side_rounds for the rows of class `cls_id` (skip = untouched rows and rows of other classes)
void side_rounds(const Side &s, int64_t n, const int32_t *list, int stride, int w, int ng, int per_slice, int pad_to, int inl_max,
                 const std::vector<int32_t> &class_rows, std::vector<int32_t> &desc, std::vector<int32_t> &iters) {
    // the class's touched rows (increasing) by slice count (a power of two up to ng)
    std::vector<std::vector<int32_t>> by_g;
    int lg_max = 0;
    while ((1 << lg_max) < ng) lg_max++;
    by_g.resize((size_t)lg_max + 1);
    for (int32_t r : class_rows) {
        const int64_t c = s.cnt[(size_t)r];
        const int64_t need = std::max<int64_t>(1, cdiv(c, per_slice));
        int lg = 0;
        while (((int64_t)1 << lg) < need) lg++;
        by_g[(size_t)std::min(lg, lg_max)].push_back(r);
    }
    size_t total_rounds = 0;
    for (int lg = lg_max; lg >= 0; lg--) total_rounds += (size_t)cdiv((int64_t)by_g[(size_t)lg].size(), ng >> lg);
    desc.reserve(desc.size() + (total_rounds + (size_t)pad_to) * (size_t)ng * 8);
    iters.reserve(iters.size() + total_rounds + (size_t)pad_to);
    const size_t slot_w = 8;
    for (int lg = lg_max; lg >= 0; lg--) {
        std::vector<int32_t> &rows = by_g[(size_t)lg];
        if (rows.empty()) continue;
        const int g = 1 << lg;
        std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return s.cnt[(size_t)a] > s.cnt[(size_t)b]; });
        const int per_round = ng / g;
        const int64_t n_rounds = cdiv((int64_t)rows.size(), per_round);
        const size_t base = desc.size(), ibase = iters.size();
        desc.resize(base + (size_t)n_rounds * ng * slot_w, 0);
        iters.resize(ibase + (size_t)n_rounds, 0);
        for (int64_t rr = 0; rr < n_rounds; rr++)
            for (int sl = 0; sl < ng; sl++) {
                int32_t *d = desc.data() + base + ((size_t)rr * ng + sl) * slot_w;
                d[0] = -1;
                d[1] = (g & 31) << 1;   // idle slots still tell the round's slice count (sync decision); 32 slices: 0
            }
        for (size_t i = 0; i < rows.size(); i++) {
            const int64_t row = rows[i], rnd = (int64_t)i / per_round, first = ((int64_t)i % per_round) * g;
            const int64_t c = s.cnt[(size_t)row], p0 = s.ptr[(size_t)row], p1 = s.ptr[(size_t)row + 1];
            const int64_t sl = std::max<int64_t>(cdiv(c, g), 1);
            for (int k = 0; k < g; k++) {
                const int64_t j0 = std::min(p0 + k * sl, p1), j1 = std::min(j0 + sl, p1), m = j1 - j0;
                const int64_t mode = m <= inl_max ? m : kModeList;
                const int64_t meta = (k == 0 ? 1 : 0) | ((int64_t)(g & 31) << 1) | (mode << 6) | (c << 9);
                int32_t *d = desc.data() + base + ((size_t)rnd * ng + (size_t)(first + k)) * slot_w;
                d[0] = (int32_t)row;
                d[1] = (int32_t)meta;
                if (n > 0) {
                    const bool inl = mode != kModeList;
                    d[2] = inl ? 0 : (int32_t)j0;
                    d[3] = inl ? 0 : (int32_t)j1;
                    for (int q = 0; q < inl_max; q++) {
                        const int64_t jq = std::min(j0 + q, n - 1);
                        if (inl && m > q)
                            for (int cc = 0; cc < w; cc++) d[2 + q * w + cc] = list[(size_t)jq * stride + cc];
                    }
                    // (a list slice's leading interactions ride in the descriptor's spare words: plan.py)
                    if (!inl && inl_max > 0)
                        for (int q = 0; q < (8 - 4) / w && q < m; q++)
                            for (int cc = 0; cc < w; cc++) d[4 + q * w + cc] = list[(size_t)(j0 + q) * stride + cc];
                }
                int32_t &it = iters[ibase + (size_t)rnd];
                it = std::max(it, (int32_t)m);
            }
        }
    }
}

// the rounds of ONE class (into arrays of their own), padded to a multiple of pad_to rounds with idle rounds
void class_rounds(const Side &s, int64_t n, const int32_t *list, int stride, int w, int ng, int per_slice, int pad_to, int inl_max,
                  const std::vector<int32_t> &class_rows, std::vector<int32_t> &d, std::vector<int32_t> &it, int snake) {
    side_rounds(s, n, list, stride, w, ng, per_slice, pad_to, inl_max, class_rows, d, it);
    if (snake > 0 && it.size() > 1) {
        // co-residency order (plan.py: _side_rounds): heaviest rounds first, every other row of `snake` rounds reversed
        const size_t nr = it.size(), slot = (size_t)ng * 8;
        std::vector<int32_t> perm(nr);
        for (size_t i = 0; i < nr; i++) perm[i] = (int32_t)i;
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return it[(size_t)a] > it[(size_t)b]; });
        for (size_t r = 1; r * (size_t)snake < nr; r += 2)
            std::reverse(perm.begin() + (ptrdiff_t)(r * snake), perm.begin() + (ptrdiff_t)std::min(nr, (r + 1) * (size_t)snake));
        std::vector<int32_t> d2(d.size()), it2(nr);
        for (size_t i = 0; i < nr; i++) {
            std::memcpy(d2.data() + i * slot, d.data() + (size_t)perm[i] * slot, slot * 4);
            it2[i] = it[(size_t)perm[i]];
        }
        d.swap(d2);
        it.swap(it2);
    }
    const int64_t have = (int64_t)it.size();
    const int64_t pad = ((-have) % pad_to + pad_to) % pad_to;
    for (int64_t p = 0; p < pad; p++) {
        for (int sl = 0; sl < ng; sl++) {
            const int32_t slot[8] = {-1, 1 << 1, 0, 0, 0, 0, 0, 0};
            d.insert(d.end(), slot, slot + 8);
        }
        it.push_back(0);
    }
}

InvPrefHostPlan *build(const int64_t *users, const int64_t *items, const float *scores, int64_t n, int64_t U, int64_t I,
                       const InvPrefPlanParams &p, int threads) {
    if (n < 0 || U <= 0 || I <= 0 || p.n_classes < 1 || p.n_classes > 8 || p.per_slice < 1 || p.item_per_slice < 1 ||
        p.rounds_per_task < 1 || p.item_rounds_per_task < 1 || p.rows_per_stream_task < 1 ||
        (p.lanes_per_group != 16 && p.lanes_per_group != 32 && p.lanes_per_group != 64) || (n > 0 && (!users || !items || !scores)))
        return nullptr;
    const bool timing = std::getenv("INVPREF_PLAN_TIMING") != nullptr;   // phase times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[invpref_plan] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int ng = kThreads / p.lanes_per_group, ncls = p.n_classes;
    // a big minibatch spreads its pieces (validation, the two sorts, then 2 x n_classes round lists) over threads
    const int nt = n >= (1 << 18) ? std::max(1, threads) : 1;
    InvPrefHostPlan *hp = new InvPrefHostPlan();
    std::vector<int32_t> &uit = hp->arr[2], &sr = hp->arr[5];
    // the sorted lists are written where the sort places each interaction: user_list [n][4] = (item, position, label bits,
    // slot), item_list [n][2] = (user, slot); rec_slot[position] = the interaction's slot = its index in the item order
    int32_t *const ul = hp->alloc(3, (size_t)n * 4), *const il = hp->alloc(4, (size_t)n * 2);
    int32_t *const ps = hp->alloc(6, (size_t)n);
    lap("allocate lists");
    Side us, is;
    const int half = std::max(1, nt / 2);
    bool ok[2] = {true, true};
    parallel_for(2, nt, [&](int side) {
        if (side == 0) {
            int32_t *L = ul;
            ok[0] = sort_side(users, n, U, us, half, [&](int64_t i, int64_t j) {
                int32_t *d = L + (size_t)j * 4;
                d[0] = (int32_t)items[i];
                d[1] = (int32_t)i;
                std::memcpy(d + 2, scores + i, 4);   // (word 3, the slot: below, once the item side has sorted)
            });
        } else {
            int32_t *L = il, *P = ps;
            ok[1] = sort_side(items, n, I, is, half, [&](int64_t i, int64_t j) {
                L[(size_t)j * 2] = (int32_t)users[i];
                L[(size_t)j * 2 + 1] = (int32_t)j;
                P[(size_t)i] = (int32_t)j;
            });
        }
    });
    if (!ok[0] || !ok[1]) {
        delete hp;
        return nullptr;
    }
    lap("sort both sides");
    {   // the slot of every entry of the user list (the item side's sort has placed them all by now)
        const int chunks = nt > 1 ? nt * 4 : 1;
        parallel_for(chunks, nt, [&](int c) {
            const int64_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
            for (int64_t k = lo; k < hi; k++) ul[(size_t)k * 4 + 3] = ps[(size_t)ul[(size_t)k * 4 + 1]];
        });
    }
    lap("slots into the user list");
    // touched rows of every class, in increasing order
    std::vector<std::vector<int32_t>> urows((size_t)ncls), irows((size_t)ncls);
    for (int64_t r = 0; r < U; r++)
        if (us.cnt[(size_t)r]) urows[(size_t)((r >> kClassShift) % ncls)].push_back((int32_t)r);
    for (int64_t r = 0; r < I; r++)
        if (is.cnt[(size_t)r]) irows[(size_t)((r >> kClassShift) % ncls)].push_back((int32_t)r);
    lap("class rows");
    std::vector<std::vector<int32_t>> cd((size_t)(2 * ncls)), cit((size_t)(2 * ncls));
    parallel_for(2 * ncls, nt, [&](int q) {
        const int c = q >> 1;
        if ((q & 1) == 0)
            class_rounds(us, n, ul, 4, 3, ng, p.per_slice, p.rounds_per_task, 2, urows[(size_t)c], cd[(size_t)q], cit[(size_t)q], p.snake_user);
        else
            class_rounds(is, n, il, 2, 2, ng, p.item_per_slice, p.item_rounds_per_task, p.push ? 0 : 3, irows[(size_t)c],
                         cd[(size_t)q], cit[(size_t)q], 0);
    });
    lap("rounds of the classes");
    int32_t cls[8][8];
    std::memset(cls, 0, sizeof(cls));
    // untouched rows: streamed, no job
    std::vector<int32_t> stream_u, stream_i;
    for (int64_t r = 0; r < U; r++)
        if (us.cnt[(size_t)r] == 0 && r >= p.user_lo && r < p.user_hi) stream_u.push_back((int32_t)r);
    for (int64_t r = 0; r < I; r++)
        if (is.cnt[(size_t)r] == 0) stream_i.push_back((int32_t)r);
    std::vector<std::vector<int32_t>> s1((size_t)ncls), s2((size_t)ncls);
    int32_t ub = 0, ib = 0;
    std::vector<int32_t> n_ur((size_t)ncls), n_ir((size_t)ncls);
    // every class's untouched rows: users, then items (a few untouched item rows are not worth one tiny task per class:
    // class 0 streams them all then)
    std::vector<std::vector<int32_t>> srows((size_t)ncls);
    for (int32_t r : stream_u) srows[(size_t)((r >> kClassShift) % ncls)].push_back(r);
    for (int32_t r : stream_i) srows[(size_t)(stream_i.size() > 8 * 64 ? (r >> kClassShift) % ncls : 0)].push_back(r | kItemBit);
    for (int c = 0; c < ncls; c++) {
        n_ur[(size_t)c] = (int32_t)cit[(size_t)(2 * c)].size();
        n_ir[(size_t)c] = (int32_t)cit[(size_t)(2 * c + 1)].size();
        uit.insert(uit.end(), cit[(size_t)(2 * c)].begin(), cit[(size_t)(2 * c)].end());
        const std::vector<int32_t> &rows = srows[(size_t)c];
        int64_t k = (int64_t)std::nearbyint(p.stream_split * (double)rows.size());   // (Python's round(): ties to even)
        if (p.fill_cap) {   // (per class: the grid is n_classes x the longest class)
            const int64_t room = p.fill_cap / ncls - cdiv(n_ur[(size_t)c], p.rounds_per_task);
            k = std::min<int64_t>((int64_t)rows.size(), std::max<int64_t>(k, room * p.rows_per_stream_task));
        }
        k = std::max<int64_t>(0, std::min<int64_t>(k, (int64_t)rows.size()));
        // inside each launch's share: item rows first, user rows last
        for (int li = 0; li < 2; li++) {
            const int64_t a = li == 0 ? 0 : k, b = li == 0 ? k : (int64_t)rows.size();
            std::vector<int32_t> &dst = li == 0 ? s1[(size_t)c] : s2[(size_t)c];
            for (int64_t q = a; q < b; q++)
                if (rows[(size_t)q] & kItemBit) dst.push_back(rows[(size_t)q]);
            for (int64_t q = a; q < b; q++)
                if (!(rows[(size_t)q] & kItemBit)) dst.push_back(rows[(size_t)q]);
        }
    }
    for (int c = 0; c < ncls; c++) {
        cls[c][0] = ub; cls[c][1] = n_ur[(size_t)c]; ub += n_ur[(size_t)c];
        cls[c][4] = ib; cls[c][5] = n_ir[(size_t)c]; ib += n_ir[(size_t)c];
    }
    int32_t sb = 0;
    for (int c = 0; c < ncls; c++) { cls[c][2] = sb; cls[c][3] = (int32_t)s1[(size_t)c].size(); sb += cls[c][3]; }
    for (int c = 0; c < ncls; c++) { cls[c][6] = sb; cls[c][7] = (int32_t)s2[(size_t)c].size(); sb += cls[c][7]; }
    {   // the classes' descriptor rounds, one after the other (copied on threads)
        std::vector<size_t> off((size_t)(2 * ncls));
        size_t tot[2] = {0, 0};
        for (int q = 0; q < 2 * ncls; q++) { off[(size_t)q] = tot[q & 1]; tot[q & 1] += cd[(size_t)q].size(); }
        int32_t *const dst[2] = {hp->alloc(0, tot[0]), hp->alloc(1, tot[1])};
        parallel_for(2 * ncls, nt, [&](int q) {
            if (!cd[(size_t)q].empty()) std::memcpy(dst[q & 1] + off[(size_t)q], cd[(size_t)q].data(), cd[(size_t)q].size() * 4);
            std::vector<int32_t>().swap(cd[(size_t)q]);
        });
    }
    for (int c = 0; c < ncls; c++) sr.insert(sr.end(), s1[(size_t)c].begin(), s1[(size_t)c].end());
    for (int c = 0; c < ncls; c++) sr.insert(sr.end(), s2[(size_t)c].begin(), s2[(size_t)c].end());
    lap("concatenate + stream rows");
    hp->arr[7].assign(&cls[0][0], &cls[0][0] + 64);
    return hp;
}

// ---- alt plans (plan.py: build_alt_plan, the same arrays byte for byte)
InvPrefHostPlan *build_alt(const int64_t *cu, const int64_t *ci, const float *cy, int64_t n, const int64_t *pu, const int64_t *pi,
                           int64_t n_prev, int64_t U, int64_t I, const InvPrefAltPlanParams &p) {
    const bool has_cur = n > 0, has_prev = pu != nullptr && n_prev >= 0;
    if (U <= 0 || I <= 0 || n < 0 || (p.side != 0 && p.side != 1) || p.per_slice < 1 || p.n_classes < 1 || p.n_classes > 8 ||
        p.pend_job_min < 1 || p.pend_per_slice < 1 || (p.slots != 16 && p.slots != 32) || (has_cur && (!cu || !ci || !cy)) || (has_prev && (!pu || !pi)))
        return nullptr;
    const int ng = p.slots;
    const int ncls = p.n_classes;
    const int64_t own_num = p.side == 0 ? U : I, oth_num = p.side == 0 ? I : U;
    const int64_t *own = p.side == 0 ? cu : ci, *oth = p.side == 0 ? ci : cu, *ownp = p.side == 0 ? pu : pi;
    InvPrefHostPlan *hp = new InvPrefHostPlan();
    std::vector<int32_t> &lst = hp->arr[2], &ps = hp->arr[3];
    Side S, T, Sp;
    lst.assign((size_t)n * 4, 0);
    ps.assign((size_t)n, 0);
    bool ok = sort_side(own, n, own_num, S, 1, [&](int64_t i, int64_t j) {
        int32_t *d = lst.data() + (size_t)j * 4;
        d[0] = (int32_t)oth[i];
        d[1] = (int32_t)i;
        std::memcpy(d + 2, cy + i, 4);
    });
    ok = ok && sort_side(oth, n, oth_num, T, 1, [&](int64_t i, int64_t j) { ps[(size_t)i] = (int32_t)j; });
    ok = ok && sort_side(ownp, has_prev ? n_prev : 0, own_num, Sp, 1, [&](int64_t, int64_t) {});
    if (!ok) {
        delete hp;
        return nullptr;
    }
    std::vector<int32_t> &desc = hp->arr[0], &pend = hp->arr[1], &stream = hp->arr[4];
    int32_t cls[8][4];
    std::memset(cls, 0, sizeof(cls));
    int32_t rb = 0;
    std::vector<std::vector<int32_t>> srows((size_t)ncls);
    for (int c = 0; c < ncls; c++) {
        std::vector<int32_t> arows, jrows;
        for (int64_t r = 0; r < own_num; r++) {
            if ((r >> kClassShift) % ncls != c) continue;
            if (S.cnt[(size_t)r] > 0) arows.push_back((int32_t)r);
            else if (Sp.cnt[(size_t)r] >= p.pend_job_min) jrows.push_back((int32_t)r);
            else srows[(size_t)c].push_back((int32_t)r);
        }
        std::vector<int32_t> d, it;
        if (has_cur) side_rounds(S, n, lst.data(), 4, 3, ng, p.per_slice, 1, 2, arows, d, it);
        const size_t n_a = d.size() / ((size_t)ng * 8);
        if (!jrows.empty()) side_rounds(Sp, 0, nullptr, 0, 0, ng, p.pend_per_slice, 1, 0, jrows, d, it);
        const size_t nr = d.size() / ((size_t)ng * 8);
        const size_t pbase = pend.size();
        pend.resize(pbase + nr * ng * 4, 0);
        for (size_t r = 0; r < nr; r++) {
            bool any = false;
            for (int sl = 0; sl < ng; sl++) {
                int32_t *dd = d.data() + (r * ng + sl) * 8, *pp = pend.data() + pbase + (r * ng + sl) * 4;
                const int64_t row = dd[0];
                if (row < 0) continue;
                const int gf = (dd[1] >> 1) & 31, g = gf ? gf : 32, k = sl % g;   // (32 slices are stored as 0)
                const int64_t cp = Sp.cnt[(size_t)row], p0 = Sp.ptr[(size_t)row];
                const int64_t ln = cdiv(cp, g);
                pp[0] = (int32_t)(p0 + std::min<int64_t>(k * ln, cp));
                pp[1] = (int32_t)(p0 + std::min<int64_t>((k + 1) * ln, cp));
                pp[2] = (int32_t)cp;
                any = any || cp > 0;
                if (r >= n_a) {   // a pending-only job: no interactions (mode 0, count 0), leader / slices bits kept
                    dd[1] &= 0x3f;
                    for (int q = 2; q < 8; q++) dd[q] = 0;
                }
            }
            if (any)
                for (int sl = 0; sl < ng; sl++) d[(r * ng + sl) * 8 + 1] |= (int32_t)0x80000000u;
        }
        desc.insert(desc.end(), d.begin(), d.end());
        cls[c][0] = rb;
        cls[c][1] = (int32_t)nr;
        rb += (int32_t)nr;
    }
    int32_t sb = 0;
    for (int c = 0; c < ncls; c++) {
        cls[c][2] = sb;
        cls[c][3] = (int32_t)srows[(size_t)c].size();
        sb += cls[c][3];
        for (int32_t r : srows[(size_t)c]) {
            const int32_t e[4] = {r, (int32_t)Sp.ptr[(size_t)r], (int32_t)Sp.ptr[(size_t)r + 1], (int32_t)Sp.cnt[(size_t)r]};
            stream.insert(stream.end(), e, e + 4);
        }
    }
    hp->arr[5].assign(&cls[0][0], &cls[0][0] + 32);
    return hp;
}

}  // namespace

extern "C" {

static InvPrefHostPlan *build_alt_guarded(const int64_t *cu, const int64_t *ci, const float *cy, int64_t n, const int64_t *pu,
                                          const int64_t *pi, int64_t n_prev, int64_t U, int64_t I, const InvPrefAltPlanParams *p) {
    if (!p) return nullptr;
    try {
        return build_alt(cu, ci, cy, n, pu, pi, n_prev, U, I, *p);
    } catch (...) {
        return nullptr;
    }
}

InvPrefHostPlan *invpref_alt_plan_build(const int64_t *cur_users, const int64_t *cur_items, const float *cur_scores, int64_t n,
                                        const int64_t *prev_users, const int64_t *prev_items, int64_t n_prev,
                                        int64_t user_num, int64_t item_num, const InvPrefAltPlanParams *params) {
    return build_alt_guarded(cur_users, cur_items, cur_scores, n, prev_users, prev_items, n_prev, user_num, item_num, params);
}

int invpref_alt_plan_build_many(const int64_t *users, const int64_t *items, const float *scores, const int64_t *cur_lo,
                                const int64_t *cur_n, const int64_t *prev_lo, const int64_t *prev_n, const int32_t *side,
                                int32_t count, int64_t user_num, int64_t item_num, const InvPrefAltPlanParams *params,
                                InvPrefHostPlan **out, int32_t n_threads) {
    if (!users || !items || !scores || !cur_lo || !cur_n || !prev_lo || !prev_n || !side || !params || !out || count < 0) return -1;
    int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(kMaxThreads, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::max(1, std::min(nt, (int)count));
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= count) return;
            InvPrefAltPlanParams p = *params;
            p.side = side[k];
            const bool hp = prev_n[k] >= 0;
            out[k] = build_alt_guarded(users + cur_lo[k], items + cur_lo[k], scores + cur_lo[k], cur_n[k],
                                       hp ? users + prev_lo[k] : nullptr, hp ? items + prev_lo[k] : nullptr, hp ? prev_n[k] : 0,
                                       user_num, item_num, &p);   // (catches)
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < nt; t++) th.emplace_back(work);
    } catch (...) {
    }
    work();
    for (auto &t : th) t.join();
    for (int k = 0; k < count; k++)
        if (!out[k]) return -2;
    return 0;
}

static InvPrefHostPlan *build_guarded(const int64_t *users, const int64_t *items, const float *scores, int64_t n,
                                      int64_t user_num, int64_t item_num, const InvPrefPlanParams *params, int threads) {
    if (!params) return nullptr;
    try {
        return build(users, items, scores, n, user_num, item_num, *params, threads);
    } catch (...) {
        return nullptr;
    }
}

InvPrefHostPlan *invpref_plan_build(const int64_t *users, const int64_t *items, const float *scores, int64_t n,
                                    int64_t user_num, int64_t item_num, const InvPrefPlanParams *params) {
    const int hw = (int)std::min<unsigned>(kMaxThreads, std::max(1u, std::thread::hardware_concurrency()));
    return build_guarded(users, items, scores, n, user_num, item_num, params, hw);
}

int64_t invpref_plan_array(const InvPrefHostPlan *plan, int32_t which, const int32_t **data) {
    if (!plan || which < 0 || which > 7) return -1;
    if (plan->big[which]) {
        if (data) *data = plan->big[which].get();
        return (int64_t)plan->big_n[which];
    }
    if (data) *data = plan->arr[which].data();
    return (int64_t)plan->arr[which].size();
}

void invpref_plan_free(InvPrefHostPlan *plan) { delete plan; }

int invpref_plan_row_counts(const int64_t *rows, int64_t n, int64_t n_rows, int64_t *counts) {
    if (n < 0 || n_rows <= 0 || !counts || (n > 0 && !rows)) return -1;
    const int hw = (int)std::min<unsigned>(kMaxThreads, std::max(1u, std::thread::hardware_concurrency()));
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(hw, n >> 18), (int64_t(64) << 20) / n_rows));
    const int64_t chunk = cdiv_i(std::max<int64_t>(n, 1), T);
    std::vector<std::vector<int32_t>> h((size_t)T);
    std::atomic<int> invalid{0};
    parallel_for(T, T, [&](int t) {
        h[(size_t)t].assign((size_t)n_rows, 0);
        const int64_t a = std::min(n, t * chunk), b = std::min(n, a + chunk);
        for (int64_t i = a; i < b; i++) {
            const uint64_t r = (uint64_t)rows[i];
            if (r < (uint64_t)n_rows) h[(size_t)t][(size_t)r]++;
            else invalid.store(1);
        }
    });
    if (invalid.load()) return -2;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t c = 0;
        for (int t = 0; t < T; t++) c += h[(size_t)t][(size_t)r];
        counts[r] = c;
    }
    return 0;
}

int invpref_plan_build_many(const int64_t *users, const int64_t *items, const float *scores, const int64_t *offsets,
                            int32_t count, int64_t user_num, int64_t item_num, const InvPrefPlanParams *params,
                            InvPrefHostPlan **out, int32_t n_threads) {
    if (!offsets || !params || !out || count < 0) return -1;
    int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(kMaxThreads, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::max(1, std::min(nt, (int)count));
    const int hw_all = (int)std::min<unsigned>(kMaxThreads, std::max(1u, std::thread::hardware_concurrency()));
    const int inner = std::max(1, hw_all / nt);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= count) return;
            const int64_t lo = offsets[k], n = offsets[k + 1] - lo;
            // (a large minibatch spreads over threads of its own: what the pool leaves of the machine, not all of it again)
            out[k] = build_guarded(users + lo, items + lo, scores + lo, n, user_num, item_num, params + k, inner);   // (catches)
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < nt; t++) th.emplace_back(work);
    } catch (...) {
    }
    work();
    for (auto &t : th) t.join();
    for (int k = 0; k < count; k++)
        if (!out[k]) return -2;
    return 0;
}

}  // extern "C"
