// hnsw_baseline.cpp -- CPU BASELINE of the reference's search ALGORITHM.  TEST / BENCHMARK INFRASTRUCTURE ONLY:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; guidemaker_amd/ never does.
//
// The reference searches with nmslib's HNSW index (guidemaker/core.py:451-456 `nmslib.init(..., method='hnsw')`,
// `createIndex({'M', 'indexThreadQty', 'efConstruction', 'post'})`; core.py:501-503 `setQueryTimeParams({'efSearch'})`,
// `knnQueryBatch(k, num_threads)`) with the shipped parameters M 16, efConstruction 10, post 1, ef 9
// (guidemaker/data/config_default.yaml:3-6).  nmslib 2.1.1 is a third-party dependency that is not vendored in the
// reference and not installable here (requirements.txt:31), so this file restates the PUBLISHED algorithm --
// Malkov & Yashunin, "Efficient and robust approximate nearest neighbor search using Hierarchical Navigable Small
// World graphs" (TPAMI 2018), Algorithms 1-5 -- with nmslib's conventions where they are documented:
//   * level of a new point: floor(-ln(U) / ln(M)); M links per node on the upper layers, 2 M on layer 0;
//   * neighbour selection by the heuristic of Algorithm 4 (nmslib `delaunay_type` 2, its default);
//   * the search beam is max(efSearch, k) (nmslib's generic-space HNSW);
//   * `post` = 1 / 2 (nmslib's post-processing, as its HNSW index builder does it): a SECOND graph is built over the same
//     points (fresh random levels, fresh insertion), and every point's layer-0 neighbour list becomes the UNION of its
//     lists in the two graphs -- kept whole for post = 1 (up to 4 M links), pruned back to 2 M with the selection
//     heuristic for post = 2; the upper layers and the entry point are the second graph's.  With the union the
//     notebook's recall on the Carsonella fixture is reproduced (tests/test_oracle.py: 0.98 .. 0.99 at (M 16, efC 13,
//     ef 13) where a single graph gives 0.95); restated from memory of the published source, named as such in the
//     bench output.
// Distance = nmslib `bit_hamming` on the one-hot encoding (2 x mismatching bases, core.py:379-386) computed on the
// 2-bit packed words and reported in TRUE units (mismatches), or `leven` (unit-cost edit distance, O(L^2) DP).
// It exists so that a "reference-algorithm CPU path" number -- build + query wall clock on the box's host cores and
// the distance recall of tests/GridOptimization.ipynb:175-190 against the exact result -- can be printed next to the
// GPU number even though nmslib itself is absent.  It is NOT the parity oracle (that is the exact scan, gm_oracle.c).
//
// Build: g++ -O3 -march=x86-64-v2 -mpopcnt -fopenmp -fPIC -shared -o libhnsw_baseline.so hnsw_baseline.cpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <queue>
#include <random>
#include <vector>

#include <omp.h>

namespace {

struct Index {
    int L = 0, metric = 0, M = 16, M0 = 32, efc = 10;
    int stride0 = 33;                             // words per node in links0 (M0 + 1; the merged graph of `post`: up to 4 M + 1)
    int64_t n = 0;
    std::vector<uint64_t> data;
    std::vector<int> level;                       // top layer of every node
    std::vector<int32_t> links0;                  // layer 0: n x (M0 + 1), [0] = count
    std::vector<std::vector<int32_t>> links_up;   // layers >= 1: per node, level x (M + 1)
    std::vector<std::mutex> locks;
    std::mutex top_lock;
    int64_t entry = -1;
    int max_level = -1;
};

inline int hamming(uint64_t a, uint64_t b) {
    const uint64_t x = a ^ b;
    return __builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
}

int leven(uint64_t a, uint64_t b, int L) {
    int prev[33], cur[33];
    for (int j = 0; j <= L; ++j) prev[j] = j;
    for (int i = 1; i <= L; ++i) {
        cur[0] = i;
        const unsigned ca = (unsigned)((a >> (2 * (i - 1))) & 3u);
        for (int j = 1; j <= L; ++j) {
            const unsigned cb = (unsigned)((b >> (2 * (j - 1))) & 3u);
            int v = prev[j - 1] + (ca != cb);
            v = std::min(v, prev[j] + 1);
            v = std::min(v, cur[j - 1] + 1);
            cur[j] = v;
        }
        std::memcpy(prev, cur, sizeof(int) * (L + 1));
    }
    return prev[L];
}

inline int dist(const Index& ix, uint64_t a, uint64_t b) { return ix.metric == 0 ? hamming(a, b) : leven(a, b, ix.L); }

inline int32_t* links(Index& ix, int64_t node, int lev) {
    return lev == 0 ? &ix.links0[(size_t)node * ix.stride0] : &ix.links_up[(size_t)node][(size_t)(lev - 1) * (ix.M + 1)];
}

struct Visited {
    std::vector<uint32_t> tag;
    uint32_t epoch = 0;
    void reset(int64_t n) {
        if ((int64_t)tag.size() != n) tag.assign((size_t)n, 0u);
        if (++epoch == 0u) {
            std::fill(tag.begin(), tag.end(), 0u);
            epoch = 1;
        }
    }
    bool test_and_set(int64_t i) {
        if (tag[(size_t)i] == epoch) return true;
        tag[(size_t)i] = epoch;
        return false;
    }
};

using Cand = std::pair<int, int32_t>;      // (distance, id)

// Algorithm 2: beam search on one layer; returns up to ef nearest found, as a max-heap
std::priority_queue<Cand> search_layer(Index& ix, uint64_t q, int64_t ep, int ef, int lev, Visited& vis, bool lock) {
    std::priority_queue<Cand> top;                                        // worst on top
    std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> cand;   // best on top
    vis.reset(ix.n);
    const int d0 = dist(ix, q, ix.data[(size_t)ep]);
    top.emplace(d0, (int32_t)ep);
    cand.emplace(d0, (int32_t)ep);
    vis.test_and_set(ep);
    std::vector<int32_t> nb;
    while (!cand.empty()) {
        const Cand c = cand.top();
        if (c.first > top.top().first && (int)top.size() >= ef) break;
        cand.pop();
        {
            if (lock) ix.locks[(size_t)c.second].lock();
            const int32_t* l = links(ix, c.second, lev);
            nb.assign(l + 1, l + 1 + l[0]);
            if (lock) ix.locks[(size_t)c.second].unlock();
        }
        for (int32_t v : nb) {
            if (vis.test_and_set(v)) continue;
            const int d = dist(ix, q, ix.data[(size_t)v]);
            if ((int)top.size() < ef || d < top.top().first) {
                cand.emplace(d, v);
                top.emplace(d, v);
                if ((int)top.size() > ef) top.pop();
            }
        }
    }
    return top;
}

// Algorithm 4: keep a candidate only if it is closer to the base point than to every neighbour kept so far
std::vector<int32_t> select_heuristic(Index& ix, std::vector<Cand> cands, int m) {
    std::sort(cands.begin(), cands.end());
    std::vector<int32_t> out;
    if ((int)cands.size() < m) {      // fewer candidates than links wanted: all of them are kept, nothing is pruned (nmslib /
        for (const Cand& c : cands) out.push_back(c.second);      // hnswlib `getNeighborsByHeuristic2`: with efConstruction <
        return out;                                                // M -- the shipped 10 < 16 -- a new point links to all it found)
    }
    for (const Cand& c : cands) {
        if ((int)out.size() >= m) break;
        bool keep = true;
        for (int32_t r : out)
            if (dist(ix, ix.data[(size_t)c.second], ix.data[(size_t)r]) < c.first) {
                keep = false;
                break;
            }
        if (keep) out.push_back(c.second);
    }
    return out;
}

void connect(Index& ix, int64_t node, const std::vector<int32_t>& sel, int lev) {
    const int cap = lev == 0 ? ix.M0 : ix.M;
    {
        std::lock_guard<std::mutex> g(ix.locks[(size_t)node]);
        int32_t* l = links(ix, node, lev);
        l[0] = (int32_t)sel.size();
        for (size_t i = 0; i < sel.size(); ++i) l[1 + i] = sel[i];
    }
    for (int32_t nbr : sel) {
        std::lock_guard<std::mutex> g(ix.locks[(size_t)nbr]);
        int32_t* l = links(ix, nbr, lev);
        bool present = false;
        for (int i = 0; i < l[0]; ++i) present = present || l[1 + i] == (int32_t)node;
        if (present) continue;
        if (l[0] < cap) {
            l[1 + l[0]++] = (int32_t)node;
        } else {      // shrink with the same heuristic
            std::vector<Cand> c;
            c.emplace_back(dist(ix, ix.data[(size_t)nbr], ix.data[(size_t)node]), (int32_t)node);
            for (int i = 0; i < l[0]; ++i) c.emplace_back(dist(ix, ix.data[(size_t)nbr], ix.data[(size_t)l[1 + i]]), l[1 + i]);
            const std::vector<int32_t> keep = select_heuristic(ix, c, cap);
            l[0] = (int32_t)keep.size();
            for (size_t i = 0; i < keep.size(); ++i) l[1 + i] = keep[i];
        }
    }
}

void insert(Index& ix, int64_t node, Visited& vis) {
    const int lev = ix.level[(size_t)node];
    std::unique_lock<std::mutex> top(ix.top_lock);
    if (ix.entry < 0) {
        ix.entry = node;
        ix.max_level = lev;
        return;
    }
    int64_t ep = ix.entry;
    const int max_level = ix.max_level;
    if (lev <= max_level) top.unlock();      // a new top level is linked under the lock
    const uint64_t q = ix.data[(size_t)node];
    for (int l = max_level; l > lev; --l) {      // greedy descent (ef = 1)
        bool moved = true;
        int best = dist(ix, q, ix.data[(size_t)ep]);
        while (moved) {
            moved = false;
            std::vector<int32_t> nb;
            {
                std::lock_guard<std::mutex> g(ix.locks[(size_t)ep]);
                const int32_t* ll = links(ix, ep, l);
                nb.assign(ll + 1, ll + 1 + ll[0]);
            }
            for (int32_t v : nb) {
                const int d = dist(ix, q, ix.data[(size_t)v]);
                if (d < best) {
                    best = d;
                    ep = v;
                    moved = true;
                }
            }
        }
    }
    for (int l = std::min(lev, max_level); l >= 0; --l) {
        std::priority_queue<Cand> top_c = search_layer(ix, q, ep, ix.efc, l, vis, true);
        std::vector<Cand> c;
        while (!top_c.empty()) {
            if (top_c.top().second != (int32_t)node) c.push_back(top_c.top());
            top_c.pop();
        }
        const std::vector<int32_t> sel = select_heuristic(ix, c, ix.M);
        connect(ix, node, sel, l);
        if (!c.empty()) ep = std::min_element(c.begin(), c.end())->second;
    }
    if (lev > max_level) {
        ix.entry = node;
        ix.max_level = lev;
    }
}

}  // namespace

extern "C" {

// one HNSW graph over the points (Algorithm 1 for every point, the first one serially, the rest on `threads` threads)
static Index* build_graph(const uint64_t* packed, int64_t n, int L, int metric, int M, int efc, int threads, uint64_t seed) {
    Index* ix = new Index();
    ix->L = L;
    ix->metric = metric;
    ix->M = M;
    ix->M0 = 2 * M;
    ix->stride0 = ix->M0 + 1;
    ix->efc = efc;
    ix->n = n;
    ix->data.assign(packed, packed + n);
    ix->level.resize((size_t)n);
    ix->links0.assign((size_t)n * ix->stride0, 0);
    ix->links_up.resize((size_t)n);
    std::vector<std::mutex> locks((size_t)n);
    ix->locks.swap(locks);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    const double mult = 1.0 / std::log((double)M);
    for (int64_t i = 0; i < n; ++i) {
        double u = uni(rng);
        if (u < 1e-300) u = 1e-300;
        const int lv = (int)std::floor(-std::log(u) * mult);
        ix->level[(size_t)i] = lv;
        if (lv > 0) ix->links_up[(size_t)i].assign((size_t)lv * (M + 1), 0);
    }
    {
        Visited v0;
        insert(*ix, 0, v0);
    }
#pragma omp parallel num_threads(threads)
    {
        Visited vis;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 1; i < n; ++i) insert(*ix, i, vis);
    }
    return ix;
}

// metric 0 = bit_hamming (true units), 1 = leven.  Returns an opaque handle (nullptr on bad arguments).
void* hb_build(const uint64_t* packed, int64_t n, int L, int metric, int M, int efc, int post, int threads, uint64_t seed) {
    if (!packed || n < 1 || L < 1 || L > 32 || M < 2 || efc < 1) return nullptr;
    if (threads < 1) threads = omp_get_max_threads();
    Index* ix = build_graph(packed, n, L, metric, M, efc, threads, seed);
    if (post == 1 || post == 2) {
        // post-processing: a second graph, layer-0 lists merged (see the header)
        Index* g2 = build_graph(packed, n, L, metric, M, efc, threads, seed * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull);
        const int cap = post == 1 ? 2 * ix->M0 : ix->M0;
        std::vector<int32_t> merged((size_t)n * (cap + 1), 0);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            const int32_t* a = links(*ix, i, 0);
            const int32_t* b = links(*g2, i, 0);
            std::vector<int32_t> u(b + 1, b + 1 + b[0]);
            for (int j = 0; j < a[0]; ++j)
                if (std::find(u.begin(), u.end(), a[1 + j]) == u.end()) u.push_back(a[1 + j]);
            if (post == 2) {
                std::vector<Cand> c;
                for (int32_t v : u) c.emplace_back(dist(*g2, g2->data[(size_t)i], g2->data[(size_t)v]), v);
                u = select_heuristic(*g2, c, cap);
            }
            int32_t* out = &merged[(size_t)i * (cap + 1)];
            out[0] = (int32_t)std::min<size_t>(u.size(), (size_t)cap);
            for (int j = 0; j < out[0]; ++j) out[1 + j] = u[(size_t)j];
        }
        g2->links0.swap(merged);
        g2->stride0 = cap + 1;
        delete ix;
        ix = g2;
    }
    return ix;
}

// k nearest (approximate) of every query; ids / dists: [nq * k], -1 padded, ordered by (distance, id).
int hb_query(void* handle, const uint64_t* q, int64_t nq, int k, int ef, int threads, int32_t* ids, int32_t* dists) {
    Index* ix = static_cast<Index*>(handle);
    if (!ix || !q || !ids || !dists || k < 1 || nq < 0) return -1;
    if (threads < 1) threads = omp_get_max_threads();
    const int beam = ef > k ? ef : k;
#pragma omp parallel num_threads(threads)
    {
        Visited vis;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < nq; ++i) {
            int64_t ep = ix->entry;
            int best = dist(*ix, q[i], ix->data[(size_t)ep]);
            for (int l = ix->max_level; l > 0; --l) {
                bool moved = true;
                while (moved) {
                    moved = false;
                    const int32_t* ll = links(*ix, ep, l);
                    for (int j = 0; j < ll[0]; ++j) {
                        const int d = dist(*ix, q[i], ix->data[(size_t)ll[1 + j]]);
                        if (d < best) {
                            best = d;
                            ep = ll[1 + j];
                            moved = true;
                        }
                    }
                }
            }
            std::priority_queue<Cand> found = search_layer(*ix, q[i], ep, beam, 0, vis, false);
            std::vector<Cand> c;
            while (!found.empty()) {
                c.push_back(found.top());
                found.pop();
            }
            std::sort(c.begin(), c.end());
            for (int j = 0; j < k; ++j) {
                const bool ok = j < (int)c.size();
                ids[i * k + j] = ok ? c[(size_t)j].second : -1;
                dists[i * k + j] = ok ? c[(size_t)j].first : -1;
            }
        }
    }
    return 0;
}

void hb_free(void* handle) { delete static_cast<Index*>(handle); }

int hb_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
