// gm_multi.hip -- multi-GPU entry points of the C ABI (include/gmhip.h): ONE process, n devices of one node.
//
// The path shards by candidate: the packed pool is replicated on every device (one gm_index per
// device), the candidates are cut into n contiguous, equally long (padded) shards, device g scans shard g, and ONE
// ncclAllGather per query call gives every device the whole neighbour table as one 32-bit key per neighbour
// ((distance << 27) | id, 0xffffffff = none); device 0's copy goes back to the caller.  No reduction, no second
// exchange.  RCCL is loaded lazily (dlopen librccl.so) so that single-GPU users of libgmhip.so do not pay for it;
// one ncclCommInitAll communicator per gm_multi.
//
// Host threading: the pigeonhole-partitioned scan waits for its stream inside gm_knn_query_dev, so each device's
// scan is driven by its own std::thread ("one host thread per handle" in the header); the collective is enqueued from
// the calling thread inside ncclGroupStart / ncclGroupEnd.
//
// Test mode (GMHIP_TEST_VIRTUAL_DEVICES=1 in the environment): the device list may name one ordinal several times --
// n LOGICAL shards with their own index, stream, host thread, padding and offsets on one physical GPU -- and the
// all-gather is n^2 device-to-device copies in place of ncclAllGather (a communicator cannot hold one device twice).
// Everything else of the n > 1 path -- the fan-out, the shard arithmetic, the key packing, the unpack on device 0 --
// is the code a real node runs; it lets a one-GPU box execute it (tests/test_gpu_parity.py).
//
// (The Python package drives multi-GPU as one process per GPU over torch.distributed instead --
// guidemaker_amd/distributed.py; this entry serves hosts without torch.  Do not drive RCCL through both in one
// process: PyTorch ships its own copy of librccl.so.)
#include "gm_host.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <thread>

#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // a process must not hold two copies of RCCL (PyTorch ships its own librccl.so next to libtorch_hip.so): take the
    // copy that is already loaded if there is one, load the system's only otherwise; never export its symbols
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (r.lib) break;
    }
    if (!r.lib) {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
    }
    if (!r.lib) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
    return r;
}

}  // namespace

struct gm_multi {
    int n = 0;
    std::vector<int> devices;
    std::vector<gm_index*> idx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    bool have_comm = false;
    bool virtual_devices = false;      // test mode: an ordinal occurs more than once (no communicator; copies instead)
    // per-device workspaces (grow-only): packed shard, shard results, shard keys, gathered keys
    std::vector<void*> d_q, d_ids, d_dists, d_counts, d_keys, d_all;
    int64_t cap_per = 0, cap_k = 0;
    // device 0: the unpacked full table
    void *d_full_ids = nullptr, *d_full_dists = nullptr, *d_full_counts = nullptr;
};

namespace {

void multi_free_buffers(gm_multi* m) {
    for (int g = 0; g < m->n; ++g) {
        (void)hipSetDevice(m->devices[g]);
        for (std::vector<void*>* v : {&m->d_q, &m->d_ids, &m->d_dists, &m->d_counts, &m->d_keys, &m->d_all})
            if ((*v)[g]) {
                (void)hipFree((*v)[g]);
                (*v)[g] = nullptr;
            }
    }
    if (m->n > 0) {
        (void)hipSetDevice(m->devices[0]);
        for (void** p : {&m->d_full_ids, &m->d_full_dists, &m->d_full_counts})
            if (*p) {
                (void)hipFree(*p);
                *p = nullptr;
            }
    }
    m->cap_per = m->cap_k = 0;
}

int multi_ensure(gm_multi* m, int64_t per, int k) {
    if (per <= m->cap_per && k <= m->cap_k) return GM_OK;
    multi_free_buffers(m);
    const int64_t n = m->n;
    for (int g = 0; g < m->n; ++g) {
        HIP_TRY(hipSetDevice(m->devices[g]));
        if (hipMalloc(&m->d_q[g], sizeof(uint64_t) * per) != hipSuccess ||
            hipMalloc(&m->d_ids[g], sizeof(int32_t) * per * k) != hipSuccess ||
            hipMalloc(&m->d_dists[g], sizeof(int32_t) * per * k) != hipSuccess ||
            hipMalloc(&m->d_counts[g], sizeof(int32_t) * per) != hipSuccess ||
            hipMalloc(&m->d_keys[g], sizeof(uint32_t) * per * k) != hipSuccess ||
            hipMalloc(&m->d_all[g], sizeof(uint32_t) * n * per * k) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "gm_multi: hipMalloc(shard workspaces) failed");
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (hipMalloc(&m->d_full_ids, sizeof(int32_t) * n * per * k) != hipSuccess ||
        hipMalloc(&m->d_full_dists, sizeof(int32_t) * n * per * k) != hipSuccess ||
        hipMalloc(&m->d_full_counts, sizeof(int32_t) * n * per) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "gm_multi: hipMalloc(gathered table) failed");
    m->cap_per = per;
    m->cap_k = k;
    return GM_OK;
}

// Every device's `send` block (count 32-bit words) to every device's `recv` buffer, block g at offset g * count:
// one grouped ncclAllGather over the devices' communicators -- or, for logical shards on one GPU (test mode), plain
// device-to-device copies on the receiving shard's stream.  The senders' streams have been synchronised by the caller.
int multi_all_gather(gm_multi* m, const std::vector<void*>& send, const std::vector<void*>& recv, size_t count) {
    const int n = m->n;
    if (m->have_comm) {
        Rccl& r = rccl();
        ncclResult_t e = r.GroupStart();
        for (int g = 0; g < n && e == ncclSuccess; ++g)
            e = r.AllGather(send[(size_t)g], recv[(size_t)g], count, ncclUint32, m->comms[(size_t)g], m->streams[(size_t)g]);
        const ncclResult_t e2 = r.GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) return gm_fail(GM_ERR_DEVICE, std::string("ncclAllGather: ") + r.GetErrorString(e));
        return GM_OK;
    }
    for (int g = 0; g < n; ++g) {
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        for (int h = 0; h < n; ++h)
            HIP_TRY(hipMemcpyAsync(static_cast<uint32_t*>(recv[(size_t)g]) + (size_t)h * count, send[(size_t)h], sizeof(uint32_t) * count,
                                   hipMemcpyDeviceToDevice, m->streams[(size_t)g]));
    }
    return GM_OK;
}

// run work(g) for every shard, each on a host thread of its own (the scans wait for their streams inside the call)
template <class F>
int multi_fan_out(gm_multi* m, F&& work) {
    const int n = m->n;
    std::vector<int> rcs((size_t)n, GM_OK);
    std::vector<std::string> msgs((size_t)n);
    auto run = [&](int g) {
        const int r = work(g);
        rcs[(size_t)g] = r;
        if (r != GM_OK) msgs[(size_t)g] = gm_last_error();      // (thread-local in the worker)
    };
    if (n == 1) {
        run(0);
    } else {
        std::vector<std::thread> th;
        for (int g = 0; g < n; ++g) th.emplace_back(run, g);
        for (auto& t : th) t.join();
    }
    for (int g = 0; g < n; ++g)
        if (rcs[(size_t)g] != GM_OK)
            return gm_fail(rcs[(size_t)g], "device " + std::to_string(m->devices[(size_t)g]) + " (shard " + std::to_string(g) + "): " + msgs[(size_t)g]);
    return GM_OK;
}

}  // namespace

extern "C" {

int gm_multi_create(const uint64_t* packed, int64_t n_db, int guidelen, int metric, const int* devices, int n_devices,
                    gm_multi** out) {
    GM_ENTER();
    if (!out) return gm_fail(GM_ERR_ARG, "gm_multi_create: null out");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return gm_fail(GM_ERR_ARG, "gm_multi_create: need 1..64 devices");
    bool dup = false;
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b) dup = dup || devices[a] == devices[b];
    {
        const char* e = std::getenv("GMHIP_TEST_VIRTUAL_DEVICES");
        if (dup && !(e && e[0] == '1')) return gm_fail(GM_ERR_ARG, "gm_multi_create: duplicate device ordinal");
    }
    gm_multi* m = new gm_multi();
    m->virtual_devices = dup;
    m->n = n_devices;
    m->devices.assign(devices, devices + n_devices);
    m->idx.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->comms.assign(n_devices, nullptr);
    for (std::vector<void*>* v : {&m->d_q, &m->d_ids, &m->d_dists, &m->d_counts, &m->d_keys, &m->d_all}) v->assign(n_devices, nullptr);
    int rc = GM_OK;
    for (int g = 0; g < n_devices && rc == GM_OK; ++g) {      // replicate the pool: one index per device
        rc = gm_index_create(packed, n_db, guidelen, metric, devices[g], &m->idx[g]);
        if (rc == GM_OK && (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreate(&m->streams[g]) != hipSuccess))
            rc = gm_fail(GM_ERR_DEVICE, "gm_multi_create: hipStreamCreate failed");
    }
    if (rc == GM_OK && !m->virtual_devices) {
        Rccl& r = rccl();
        if (r.ok) {
            const ncclResult_t e = r.CommInitAll(m->comms.data(), n_devices, devices);
            if (e != ncclSuccess) rc = gm_fail(GM_ERR_DEVICE, std::string("ncclCommInitAll: ") + r.GetErrorString(e));
            else m->have_comm = true;
        } else if (n_devices > 1) {
            rc = gm_fail(GM_ERR_DEVICE, "gm_multi_create: librccl.so could not be loaded (needed for more than one device)");
        }
    }
    if (rc != GM_OK) {
        gm_multi_free(m);
        return rc;
    }
    *out = m;
    return GM_OK;
}

void gm_multi_free(gm_multi* m) {
    if (!m) return;
    multi_free_buffers(m);
    for (int g = 0; g < m->n; ++g) {
        if (m->have_comm && m->comms[g]) (void)rccl().CommDestroy(m->comms[g]);
        if (m->streams[g]) {
            (void)hipSetDevice(m->devices[g]);
            (void)hipStreamDestroy(m->streams[g]);
        }
        if (m->idx[g]) gm_index_free(m->idx[g]);
    }
    delete m;
}

int gm_multi_devices(const gm_multi* m) { return m ? m->n : 0; }

int gm_multi_has_comm(const gm_multi* m) { return m && m->have_comm ? 1 : 0; }

gm_index* gm_multi_index(gm_multi* m, int i) { return (m && i >= 0 && i < m->n) ? m->idx[i] : nullptr; }

int gm_multi_knn_query(gm_multi* m, const uint64_t* q, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts) {
    GM_ENTER();
    if (!m) return gm_fail(GM_ERR_ARG, "gm_multi_knn_query: null handle");
    if (n_q < 0 || k < 1 || k > GM_MAX_K) return gm_fail(GM_ERR_ARG, "gm_multi_knn_query: need n_q >= 0, 1 <= k <= 64");
    if (n_q == 0) return GM_OK;
    if (!q || !ids || !dists || !counts) return gm_fail(GM_ERR_ARG, "gm_multi_knn_query: null buffer");
    const int n = m->n;
    const int64_t per = (n_q + n - 1) / n;
    int rc = multi_ensure(m, per, k);
    if (rc) return rc;
    // phase 1: every device scans its shard (own host thread per device), packs its lists into keys
    rc = multi_fan_out(m, [&](int g) -> int {
        const int64_t lo = per * g < n_q ? per * g : n_q, hi = lo + per < n_q ? lo + per : n_q, cnt = hi - lo;
        hipStream_t st = m->streams[(size_t)g];
        if (hipSetDevice(m->devices[(size_t)g]) != hipSuccess) return gm_fail(GM_ERR_DEVICE, "hipSetDevice failed");
        // rows past the end of the candidate list: "no neighbour" keys
        if (hipMemsetAsync(m->d_keys[(size_t)g], 0xff, sizeof(uint32_t) * per * k, st) != hipSuccess)
            return gm_fail(GM_ERR_DEVICE, "hipMemsetAsync(keys) failed");
        if (cnt > 0) {
            int r;
            if (hipMemcpyAsync(m->d_q[(size_t)g], q + lo, sizeof(uint64_t) * cnt, hipMemcpyHostToDevice, st) != hipSuccess)
                return gm_fail(GM_ERR_DEVICE, "hipMemcpyAsync(shard) failed");
            if ((r = gm_knn_query_dev(m->idx[(size_t)g], m->d_q[(size_t)g], cnt, k, m->d_ids[(size_t)g], m->d_dists[(size_t)g], m->d_counts[(size_t)g], st))) return r;
            if ((r = gm_pack_keys_dev(m->d_ids[(size_t)g], m->d_dists[(size_t)g], cnt, k, m->d_keys[(size_t)g], st))) return r;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return gm_fail(GM_ERR_DEVICE, "shard scan failed");
        return GM_OK;
    });
    if (rc) return rc;
    // phase 2: ONE all-gather of the keys over RCCL / xGMI
    const void* gathered = m->d_keys[0];
    if (m->have_comm || n > 1) {
        if ((rc = multi_all_gather(m, m->d_keys, m->d_all, (size_t)(per * k)))) return rc;
        gathered = m->d_all[0];
    }
    // phase 3: device 0 unpacks the gathered table; back to the caller
    HIP_TRY(hipSetDevice(m->devices[0]));
    rc = gm_unpack_keys_dev(gathered, (int64_t)n * per, k, m->d_full_ids, m->d_full_dists, m->d_full_counts, m->streams[0]);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ids, m->d_full_ids, sizeof(int32_t) * n_q * k, hipMemcpyDeviceToHost, m->streams[0]));
    HIP_TRY(hipMemcpyAsync(dists, m->d_full_dists, sizeof(int32_t) * n_q * k, hipMemcpyDeviceToHost, m->streams[0]));
    HIP_TRY(hipMemcpyAsync(counts, m->d_full_counts, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost, m->streams[0]));
    for (int g = 0; g < n; ++g) {      // every device's part of the collective has finished before the buffers are reused
        HIP_TRY(hipSetDevice(m->devices[g]));
        HIP_TRY(hipStreamSynchronize(m->streams[g]));
    }
    return GM_OK;
}

// Nearest-target distance of every candidate (get_control_seqs' k = 2 batch, /root/reference/guidemaker/core.py:603: the
// path's largest query set), sharded like gm_multi_knn_query: device g scans shard g (gm_nearest_dist_dev), ONE
// all-gather of the int32 distances, device 0's copy goes back.  d1[i] = -1 only for an empty pool.
int gm_multi_nearest_dist(gm_multi* m, const uint64_t* q, int64_t n_q, int32_t* d1) {
    GM_ENTER();
    if (!m) return gm_fail(GM_ERR_ARG, "gm_multi_nearest_dist: null handle");
    if (n_q < 0) return gm_fail(GM_ERR_ARG, "gm_multi_nearest_dist: need n_q >= 0");
    if (n_q == 0) return GM_OK;
    if (!q || !d1) return gm_fail(GM_ERR_ARG, "gm_multi_nearest_dist: null buffer");
    const int n = m->n;
    const int64_t per = (n_q + n - 1) / n;
    int rc = multi_ensure(m, per, 1);      // (d_keys: the shard's distances, d_all: the gathered ones)
    if (rc) return rc;
    rc = multi_fan_out(m, [&](int g) -> int {
        const int64_t lo = per * g < n_q ? per * g : n_q, hi = lo + per < n_q ? lo + per : n_q, cnt = hi - lo;
        hipStream_t st = m->streams[(size_t)g];
        if (hipSetDevice(m->devices[(size_t)g]) != hipSuccess) return gm_fail(GM_ERR_DEVICE, "hipSetDevice failed");
        if (hipMemsetAsync(m->d_keys[(size_t)g], 0xff, sizeof(uint32_t) * per, st) != hipSuccess)      // padding rows: -1
            return gm_fail(GM_ERR_DEVICE, "hipMemsetAsync(distances) failed");
        if (cnt > 0) {
            int r;
            if (hipMemcpyAsync(m->d_q[(size_t)g], q + lo, sizeof(uint64_t) * cnt, hipMemcpyHostToDevice, st) != hipSuccess)
                return gm_fail(GM_ERR_DEVICE, "hipMemcpyAsync(shard) failed");
            if ((r = gm_nearest_dist_dev(m->idx[(size_t)g], m->d_q[(size_t)g], cnt, m->d_keys[(size_t)g], st))) return r;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return gm_fail(GM_ERR_DEVICE, "shard scan failed");
        return GM_OK;
    });
    if (rc) return rc;
    const void* gathered = m->d_keys[0];
    if (m->have_comm || n > 1) {
        if ((rc = multi_all_gather(m, m->d_keys, m->d_all, (size_t)per))) return rc;
        gathered = m->d_all[0];
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpyAsync(d1, gathered, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost, m->streams[0]));
    for (int g = 0; g < n; ++g) {
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        HIP_TRY(hipStreamSynchronize(m->streams[(size_t)g]));
    }
    return GM_OK;
}

}  // extern "C"
