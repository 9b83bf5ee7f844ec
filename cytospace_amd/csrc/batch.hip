// batch.hip -- many independent LAPs on one GPU at once (the communicator of the multi-GPU fan-out is comm.hip).
//
// CytoSPACE splits large inputs into independent square sub-LAPs ("chunks") and ships each to a worker
// process (/root/reference/cytospace/cytospace.py:430-451).  Here the chunks of a batch go through the solver's phases
// TOGETHER: problems of one size share their launches (lap_jv.hip: lap_batch_same_n -- the row reduction of every problem is the
// same whole-chip phase machine, launch L carrying round L of whatever phase each problem is in; the searches run a
// workgroup per problem), and a large batch runs as a few interleaved sub-batches on their own streams and host threads.
#include "cyto_common.h"
#include <algorithm>
#include <stdlib.h>
#include <map>
#include <thread>
#include <vector>

namespace cyto {

// cyto_lap_batch_f32 with optional row maps (rowmap[b] != NULL: cost[b] holds nu[b] distinct rows, see cyto_lap_f32_rowmap)
static int lap_batch_any_impl(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                  const int32_t *const *rowmap, const int *nu, int32_t *const *rowsol, int32_t *const *colsol, float *const *u,
                  float *const *v, double *total, cyto_lap_info *info, int *status_out, int max_concurrent, int device_id,
                  const cyto_lap_opts *opts) {
    if (nb < 0 || (nb > 0 && (!n || !cost || !ld))) return CYTO_ERR_BAD_ARG;
    if (nb == 0) return CYTO_OK;
    int rc = select_device(device_id);
    if (rc) return rc;
    const int conc = std::max(1, std::min(nb, max_concurrent > 0 ? max_concurrent : 256));
    StreamGuard guard;
    if ((rc = guard.acquire())) return rc;
    std::vector<int> st((size_t)nb, CYTO_OK);
    std::map<int, std::vector<int>> by_n;                 // size -> problems, in input order
    for (int b = 0; b < nb; b++) {
        if (n[b] <= 0 || !cost[b] || ld[b] < n[b]) st[(size_t)b] = CYTO_ERR_BAD_ARG;
        else by_n[n[b]].push_back(b);
    }
    for (auto &kv : by_n) {
        const std::vector<int> &ids = kv.second;
        for (size_t lo = 0; lo < ids.size(); lo += (size_t)conc) {
            const int cnt = (int)std::min(ids.size() - lo, (size_t)conc);
            std::vector<const float *> c((size_t)cnt);
            std::vector<int64_t> l((size_t)cnt);
            std::vector<int32_t *> rs((size_t)cnt), cs((size_t)cnt);
            std::vector<float *> uu((size_t)cnt), vv((size_t)cnt);
            std::vector<double> tot((size_t)cnt);
            std::vector<cyto_lap_info> inf((size_t)cnt);
            std::vector<int> stat((size_t)cnt, CYTO_OK), nus((size_t)cnt, 0);
            std::vector<const int32_t *> rm((size_t)cnt, nullptr);
            for (int k = 0; k < cnt; k++) {
                const int b = ids[lo + (size_t)k];
                c[(size_t)k] = cost[b]; l[(size_t)k] = ld[b];
                rs[(size_t)k] = rowsol ? rowsol[b] : nullptr; cs[(size_t)k] = colsol ? colsol[b] : nullptr;
                uu[(size_t)k] = u ? u[b] : nullptr; vv[(size_t)k] = v ? v[b] : nullptr;
                if (rowmap && rowmap[b]) { rm[(size_t)k] = rowmap[b]; nus[(size_t)k] = nu ? nu[b] : 0; }
            }
            // A large batch goes as G interleaved sub-batches, each on its own stream and host thread: the phases of a sub-batch are
            // launches for all its problems, and some of them (the cache rebuilds between searches, the waits for its slowest
            // problem) leave most of the chip idle -- another sub-batch's workgroups fill it.
            // (chunk-sized problems only: 32 problems of 20 000 rows, which never pause, took 0.19 s in four sub-batches, 0.15 s in one)
            // (more sub-batches for 8-31 problems -- 2 / 4 instead of 1 / 2 -- help a lone batch call (tools/exp/subbatch_ab.sh, gpurun_out/r04ai:
            //  8 problems 63 / 59 / 59 / 97 ms in 1 / 2 / 4 / 8 sub-batches, 20 problems 91 / 80 / 144 / 210 ms in 2 / 4 / 10 / 20, 64 problems
            //  171 / 210 / 255 ms in 4 / 8 / 16) and cost the context path, whose groups of chunks already run side by side (gpurun_out/r04aj:
            //  50 c5 chunks 0.15 -> 0.21 s, configs[3]'s 20 chunks 0.31 -> 0.38 s): beyond a handful the host threads' launches get in each
            //  other's way)
            int G = kv.first > 16384 ? 1 : (cnt >= 128 ? 8 : (cnt >= 32 ? 4 : (cnt >= 16 ? 2 : 1)));
            if (CYTO_KNOB("CYTO_SUBBATCHES").set) G = std::max(1, std::min(cnt, CYTO_KNOB("CYTO_SUBBATCHES").value));       // (developer knob, read once per process: tools/batch_chunks_bench.py)
            std::vector<int> brcs((size_t)G, CYTO_OK);
            if (G == 1) {
                brcs[0] = lap_batch_same_n(kv.first, cnt, c.data(), l.data(), cost_on_device, rs.data(), cs.data(), uu.data(), vv.data(),
                                           tot.data(), inf.data(), stat.data(), device_id, guard.s, rm.data(), nus.data(), opts);
            } else {
                std::vector<std::vector<int>> part((size_t)G);
                for (int k = 0; k < cnt; k++) part[(size_t)(k % G)].push_back(k);
                // one sub-batch: a stream of its own, its problems' pointers, the solve, its results (an int status, never an exception)
                auto run_part = [&](int g) -> int {
                    try {
                        const std::vector<int> &ks = part[(size_t)g];
                        const int m = (int)ks.size();
                        if (select_device(device_id)) return CYTO_ERR_HIP;
                        StreamGuard sg;
                        if (sg.acquire()) return CYTO_ERR_HIP;
                        std::vector<const float *> c2((size_t)m); std::vector<int64_t> l2((size_t)m);
                        std::vector<int32_t *> rs2((size_t)m), cs2((size_t)m); std::vector<float *> u2((size_t)m), v2((size_t)m);
                        std::vector<double> tot2((size_t)m); std::vector<cyto_lap_info> inf2((size_t)m);
                        std::vector<int> stat2((size_t)m, CYTO_OK), nus2((size_t)m, 0); std::vector<const int32_t *> rm2((size_t)m, nullptr);
                        for (int q = 0; q < m; q++) {
                            const size_t k = (size_t)ks[(size_t)q];
                            c2[(size_t)q] = c[k]; l2[(size_t)q] = l[k]; rs2[(size_t)q] = rs[k]; cs2[(size_t)q] = cs[k];
                            u2[(size_t)q] = uu[k]; v2[(size_t)q] = vv[k]; rm2[(size_t)q] = rm[k]; nus2[(size_t)q] = nus[k];
                        }
                        const int r = lap_batch_same_n(kv.first, m, c2.data(), l2.data(), cost_on_device, rs2.data(), cs2.data(), u2.data(),
                                                       v2.data(), tot2.data(), inf2.data(), stat2.data(), device_id, sg.s, rm2.data(),
                                                       nus2.data(), opts);
                        for (int q = 0; q < m; q++) {
                            const size_t k = (size_t)ks[(size_t)q];
                            tot[k] = tot2[(size_t)q]; inf[k] = inf2[(size_t)q]; stat[k] = stat2[(size_t)q];
                        }
                        return r;
                    } catch (const std::bad_alloc &) {
                        return CYTO_ERR_NOMEM;
                    } catch (...) {
                        return CYTO_ERR_INTERNAL;
                    }
                };
                // (no exception may cross the C ABI: a thread that cannot be created -- std::system_error -- leaves its sub-batch to
                //  this thread, after the others)
                std::vector<std::thread> th;
                std::vector<int> inline_parts;
                for (int g = 0; g < G; g++) {
                    try {
                        th.emplace_back([&, g]() { brcs[(size_t)g] = run_part(g); });
                    } catch (...) {
                        inline_parts.push_back(g);
                    }
                }
                for (auto &t : th) t.join();
                for (int g : inline_parts) brcs[(size_t)g] = run_part(g);
            }
            for (int k = 0; k < cnt; k++) {
                const int b = ids[lo + (size_t)k];
                const int brc = brcs[(size_t)(G == 1 ? 0 : k % G)];
                st[(size_t)b] = brc ? brc : stat[(size_t)k];
                if (total) total[b] = tot[(size_t)k];
                if (info) info[b] = inf[(size_t)k];
            }
        }
    }
    int first = CYTO_OK;
    for (int b = 0; b < nb; b++) {
        if (status_out) status_out[b] = st[(size_t)b];
        if (!first && st[(size_t)b]) first = st[(size_t)b];
    }
    return first;
}

// No exception may cross the C ABI: the containers above (and lap_batch_same_n's) throw std::bad_alloc on exhaustion.
int lap_batch_any(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                  const int32_t *const *rowmap, const int *nu, int32_t *const *rowsol, int32_t *const *colsol, float *const *u,
                  float *const *v, double *total, cyto_lap_info *info, int *status_out, int max_concurrent, int device_id,
                  const cyto_lap_opts *opts) {
    int rc;
    try {
        return lap_batch_any_impl(nb, n, cost, ld, cost_on_device, rowmap, nu, rowsol, colsol, u, v, total, info, status_out, max_concurrent,
                                  device_id, opts);
    } catch (const std::bad_alloc &) {
        rc = CYTO_ERR_NOMEM;
    } catch (...) {
        rc = CYTO_ERR_INTERNAL;
    }
    if (status_out) for (int b = 0; b < nb; b++) status_out[b] = rc;
    return rc;
}

}  // namespace cyto

extern "C" {

// Solve nb independent LAPs.  Arrays of per-problem pointers/sizes; outputs may be NULL like in
// cyto_lap_f32.  Problems of equal size share their launches; max_concurrent bounds how many are in flight at once (each
// holds its workspace, ~2.6 KB per row): <= 0 picks min(nb, 256), one chain per CU.  Returns the first non-zero status (all
// problems are attempted); status_out[b] (optional) receives each problem's status.
int cyto_lap_batch_f32(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                       int32_t *const *rowsol, int32_t *const *colsol, float *const *u, float *const *v, double *total,
                       cyto_lap_info *info, int *status_out, int max_concurrent, int device_id) {
    return cyto::lap_batch_any(nb, n, cost, ld, cost_on_device, nullptr, nullptr, rowsol, colsol, u, v, total, info, status_out,
                               max_concurrent, device_id);
}

// The same with kernel-selection options (NULL = the defaults): cyto_lap_opts.mode picks the solver for the whole batch.
int cyto_lap_batch_f32_opts(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                            int32_t *const *rowsol, int32_t *const *colsol, float *const *u, float *const *v, double *total,
                            cyto_lap_info *info, int *status_out, int max_concurrent, int device_id, const cyto_lap_opts *opts) {
    return cyto::lap_batch_any(nb, n, cost, ld, cost_on_device, nullptr, nullptr, rowsol, colsol, u, v, total, info, status_out,
                               max_concurrent, device_id, opts);
}

}  // extern "C"
