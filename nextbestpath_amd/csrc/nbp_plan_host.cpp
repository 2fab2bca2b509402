// Host half of a replan: the candidate loop of compute_nbp_trajectory (next_best_path/testers/nbp_planning.py:233-249)
// around generate_Dijkstra_path (next_best_path/utility/long_term_utils.py:334-418) on the lattice graph.  The
// reference runs this on the host as well (heapq + dict); here it is C++ over integer node ids because with 16 rollouts
// per GPU the Python form (0.3 ms per replan) was as long as the GPU's share of a step.  Same results as
// utility/planner_host.py (level_order_tree + choose_headings), which stays as the readable statement and the fallback for
// the one case that needs the rollout's Python random stream (a path node outside the value map).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "nbp_hip.h"

namespace {

// utils.py:160-196 in fp32 (planner_host.value_cell): cell = rint((-(p - c) - lo) * sc), round-half-even
inline int cell_of(float p, float c, float lo, float sc) {
    const float v = -(p - c);
    return (int)nearbyintf((v - lo) * sc);
}

}  // namespace

extern "C" int nbp_plan_search_host(int P, const int* idx3, const float* pos3, int E, const int* edges2,
                                    const int* edge_first, const unsigned char* mesh_hit,
                                    const unsigned char* blocked, const unsigned char* coll_mask,
                                    const unsigned char* pass_mask, const int* cand, int n_cand, int start_id,
                                    float cx, float cz, const float* out1, int V, float lo, float sc,
                                    const int* hist5, int n_hist, int check_first_edge, int max_path,
                                    int* path_nodes, int* path_heads, int* path_len, int* goal, int* new_coll2,
                                    int max_new_coll, int* n_new_coll) {
    if (P <= 0 || E < 0 || !idx3 || !pos3 || !edges2 || !edge_first || !blocked || !coll_mask || !pass_mask ||
        !out1 || !path_nodes || !path_heads || !path_len || !goal || !new_coll2 || !n_new_coll || start_id < 0 ||
        start_id >= P || max_path < 1)
        return NBP_E_ARG;
    std::vector<unsigned char> coll(coll_mask, coll_mask + E), ok(E);
    std::vector<int> parent(P), level, next, ids;
    bool tree_valid = false;
    *path_len = -1;
    *goal = -1;
    *n_new_coll = 0;
    std::vector<int> cur_nodes, cur_heads;
    int cur_len = -1;                                    // the Python loop returns the LAST path it formed (or None)

    auto build_tree = [&]() {
        // ok = [a,b] in passable_list or (not blocked and [a,b] not in collision_list)   (ref :350-360)
        for (int q = 0; q < E; ++q) ok[q] = pass_mask[q] || (!blocked[q] && !coll[q]);
        std::fill(parent.begin(), parent.end(), -2);
        parent[start_id] = -1;
        level.assign(1, start_id);
        // uniform cost + heap ordered by (cost, tuple) + came_from fixed at first discovery == BFS levels expanded in
        // increasing node id (ids are assigned in lexicographic (i,j,k) order)
        while (!level.empty()) {
            std::sort(level.begin(), level.end());
            next.clear();
            for (int u : level)
                for (int q = edge_first[u]; q < edge_first[u + 1]; ++q) {
                    const int v = edges2[2 * q + 1];
                    if (ok[q] && parent[v] == -2) {
                        parent[v] = u;
                        next.push_back(v);
                    }
                }
            level.swap(next);
        }
        tree_valid = true;
    };

    auto used = [&](int n, int h) {
        const int* t = idx3 + 3 * n;
        for (int r = 0; r < n_hist; ++r) {
            const int* hr = hist5 + 5 * r;
            if (hr[0] == t[0] && hr[1] == t[1] && hr[2] == t[2] && hr[3] == 2 && hr[4] == h) return true;
        }
        return false;
    };

    for (int c = 0; c < n_cand; ++c) {
        const int ci = cand[c];
        if (ci < 0 || ci >= P) return NBP_E_ARG;
        if (!tree_valid) build_tree();
        if (parent[ci] == -2) {
            cur_len = -1;
            continue;
        }
        ids.clear();
        for (int u = ci; u >= 0; u = parent[u]) ids.push_back(u);
        std::reverse(ids.begin(), ids.end());
        // heading per node (ref :390-413): best-valued heading not used at that node yet; elevation index 2
        cur_nodes.clear();
        cur_heads.clear();
        for (int n : ids) {
            const int g0 = cell_of(pos3[3 * n + 2], cz, lo, sc), g1 = cell_of(pos3[3 * n + 0], cx, lo, sc);
            if (g0 < 0 || g0 >= V || g1 < 0 || g1 >= V) {      // random heading: drawn from the caller's Python stream
                *path_len = -2;
                *n_new_coll = 0;
                return 0;
            }
            int order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            float key[8];
            for (int h = 0; h < 8; ++h) key[h] = -out1[((size_t)h * V + g0) * V + g1];
            std::stable_sort(order, order + 8, [&](int a, int b) {          // numpy stable argsort, NaN last
                return key[a] < key[b] || (std::isnan(key[b]) && !std::isnan(key[a]));
            });
            int h = order[7];
            for (int k = 0; k < 8; ++k) {
                h = order[k];
                if (!used(n, h)) break;
            }
            cur_nodes.push_back(n);
            cur_heads.push_back(h);
        }
        cur_len = (int)ids.size() - 1;                      // the first node (the camera's own) is dropped (ref :416)
        if (cur_len > 0) {
            const int a = ids[0], b = ids[1];
            int q = -1;
            for (int e = edge_first[a]; e < edge_first[a + 1]; ++e)
                if (edges2[2 * e + 1] == b) q = e;
            if (q < 0) return NBP_E_ARG;
            if (!check_first_edge || !(mesh_hit && mesh_hit[q])) {
                *goal = ci;
                break;
            }
            // the first edge crosses the real mesh: both directions join the collision list, the tree is rebuilt
            if (*n_new_coll >= max_new_coll) return NBP_E_ARG;
            new_coll2[2 * *n_new_coll] = a;
            new_coll2[2 * *n_new_coll + 1] = b;
            ++*n_new_coll;
            coll[q] = 1;
            for (int e = edge_first[b]; e < edge_first[b + 1]; ++e)
                if (edges2[2 * e + 1] == a) coll[e] = 1;
            tree_valid = false;
        }
    }
    if (cur_len > max_path) return NBP_E_ARG;
    *path_len = cur_len;
    for (int k = 0; k < cur_len; ++k) {
        path_nodes[k] = cur_nodes[k + 1];
        path_heads[k] = cur_heads[k + 1];
    }
    return 0;
}
