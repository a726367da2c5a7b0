#!/usr/bin/env python3
"""CPU model of a design NOT yet in the kernel: hnsw_beam_kernel's runner-up selection replaced by a sorted cache of
the T best candidates.  The model replays search_layer (hnsw/block_based/index.rs:212-287: min-heap of candidates keyed
(distance asc, id desc), bounded working set of ef) twice on a k-NN graph — once with the heap, once with "unsorted candidate
set + cache" under the invariant `cache = the |cache| smallest candidates` — and checks that both pop the same nodes in the same
order; it reports how often the cache runs empty (= a full selection, today's cost on EVERY step).
usage: sim_candidate_cache.py [--n 4000] [--d 16] [--deg 16] [--ef 200] [--T 8] [--queries 50]"""
import argparse, heapq
import numpy as np


def knn_graph(x, deg, rng):
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nb = np.argsort(d2, axis=1)[:, :deg]
    adj = [list(map(int, r)) for r in nb]
    for i in range(len(adj)):          # a few long links, symmetric-ish, like the upper layers' effect
        j = int(rng.integers(len(adj)))
        if j != i and j not in adj[i]:
            adj[i][-1] = j
    return adj


def search(x, adj, q, ep, ef, T):
    dist = lambda i: float(np.float32(((x[i] - q) ** 2).sum()))
    key = lambda i: (dist(i), -i)      # pop order: smallest distance, LARGEST id among equals
    visited = {ep}
    # ---- reference: heap of candidates, W = ef smallest (distance, id)
    cand = [key(ep)]
    W = [(-dist(ep), -ep)]             # max-heap on (distance, id)
    pops_ref = []
    # ---- model: candidate set C (unordered) + cache (sorted list, <= T)
    C = {ep: key(ep)}
    cache = []                         # sorted keys; invariant: the len(cache) smallest of C
    pops_model, rebuilds, steps = [], 0, 0
    while cand:
        kd, nid = heapq.heappop(cand)
        node = -nid
        furthest = -W[0][0]
        # model pop (must name the same node)
        if not cache:
            rebuilds += 1
            cache = sorted(C.values())[:T]
        mk = cache.pop(0)
        mnode = -mk[1]
        del C[mnode]
        pops_ref.append(node); pops_model.append(mnode)
        if kd > furthest:
            break
        steps += 1
        for nb in adj[node]:
            if nb in visited:
                continue
            visited.add(nb)
            dn = dist(nb)
            furthest = -W[0][0]
            if dn < furthest or len(W) < ef:
                heapq.heappush(cand, key(nb))
                heapq.heappush(W, (-dn, -nb))
                if len(W) > ef:
                    heapq.heappop(W)
                # model insert
                k = key(nb)
                C[nb] = k
                if cache and k < cache[-1]:
                    cache.append(k); cache.sort()
                    if len(cache) > T:
                        cache.pop()        # the evicted entry stays in C
    assert pops_ref == pops_model, "pop order differs"
    return steps, rebuilds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4000); ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--deg", type=int, default=16); ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--T", type=int, default=8); ap.add_argument("--queries", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1); ap.add_argument("--integer", action="store_true", help="rounded coordinates: many exact distance ties")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    x = rng.standard_normal((a.n, a.d)).astype(np.float32)
    if a.integer:
        x = np.round(x * 3).astype(np.float32)
    adj = knn_graph(x, a.deg, rng)
    tot_s = tot_r = 0
    for _ in range(a.queries):
        q = rng.standard_normal(a.d).astype(np.float32)
        if a.integer:
            q = np.round(q * 3).astype(np.float32)
        s, r = search(x, adj, q, int(rng.integers(a.n)), a.ef, a.T)
        tot_s += s; tot_r += r
    print("T=%d ef=%d: %d steps, %d full selections (%.1f %% of the steps), same pop order on %d queries"
          % (a.T, a.ef, tot_s, tot_r, 100.0 * tot_r / max(tot_s, 1), a.queries))


if __name__ == "__main__":
    main()
