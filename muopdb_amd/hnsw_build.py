"""HNSW construction, host-side graph logic: the parts of
`HnswBuilder` (rs/index/src/hnsw/builder.rs) that are pure integer / ordering work on the graph —
the BFS renumbering of `reindex` (:100-218), `Layer::reindex` (:24-45), the level draw of
`get_random_layer` (:332-337) — restated in numpy / plain Python.  The distance-heavy part of construction
(`insert` :221-305: search_layer per level + select_neighbors_heuristic :339-375) runs on the GPU:
muopdb_amd.build.insert_hnsw drives the library's traversal kernels batch by batch.

A layer is `{point_id: [(neighbour_id, distance), ...]}` (the reference's `Layer { edges: HashMap<u32, Vec<PointAndDistance>> }`),
layer 0 first.  Pinned by the reference's own known answers K11 (builder.rs:460-620) in tests/test_oracle_kat.py.
"""
from collections import deque

import numpy as np


def layer_reindex(layer, id_mapping):
    """Layer::reindex (builder.rs:24-45): rename every point and every edge target through id_mapping."""
    return {int(id_mapping[p]): [(int(id_mapping[e]), d) for e, d in edges] for p, edges in layer.items()}


def reindex_layer(layer, assigned_ids, current_id, vector_length):
    """HnswBuilder::reindex_layer (builder.rs:100-147): BFS from every unvisited point in ascending id order; a point's
    edges are visited nearest first (stable sort by distance, done IN PLACE like the reference); ids are handed out in
    first-touch order.  Returns the next free id."""
    visited = np.zeros(vector_length, bool)
    for e in sorted(layer.keys()):
        if visited[e]:
            continue
        queue = deque([e])
        if assigned_ids[e] < 0:
            assigned_ids[e] = current_id
            current_id += 1
        while queue:
            node = queue.popleft()
            visited[node] = True
            edges = layer.get(node)
            if edges is not None:
                edges.sort(key=lambda x: x[1])  # sort_by_key is stable, so is list.sort
                for pid, _ in edges:
                    if visited[pid]:
                        continue
                    queue.append(pid)
                    if assigned_ids[pid] < 0:
                        assigned_ids[pid] = current_id
                        current_id += 1
                    visited[pid] = True
    return current_id


def get_reassigned_ids(layers, vector_length):
    """get_reassigned_ids (builder.rs:150-165): top layer first, so upper-layer points get the smallest ids."""
    assigned = np.full(vector_length, -1, np.int64)
    cur = 0
    for layer in reversed(layers):
        cur = reindex_layer(layer, assigned, cur, vector_length)
    return assigned


def reindex(layers, entry_points, doc_id_mapping, vectors):
    """HnswBuilder::reindex (builder.rs:170-218): returns (layers, entry_points, doc_id_mapping, vectors, assigned_ids) after
    the renumbering — connected points get nearby ids, vectors are permuted accordingly.  Every point of the vector
    storage must appear in layer 0 (the builder inserts every vector there), as in the reference (it indexes
    doc_id_mapping with the assigned id)."""
    n = len(doc_id_mapping)
    assigned = get_reassigned_ids(layers, n)
    if np.any(assigned < 0):
        raise ValueError("reindex: a point of the vector storage is in no layer")
    new_layers = [layer_reindex(layer, assigned) for layer in layers]
    new_docs = list(doc_id_mapping)
    for i, doc in enumerate(doc_id_mapping):
        new_docs[int(assigned[i])] = doc
    new_entry = sorted(int(assigned[e]) for e in entry_points)
    reverse = np.empty(n, np.int64)
    reverse[assigned] = np.arange(n)
    v = np.asarray(vectors)
    return new_layers, new_entry, new_docs, v[reverse], assigned


def get_random_layer(rng, max_neighbors, max_layer):
    """get_random_layer (builder.rs:332-337): floor(-ln(u) * 1/ln(max_neighbors)) capped at max_layer (u uniform in (0,1])."""
    u = 1.0 - rng.random()
    return int(min(np.floor(-np.log(u) / np.log(max_neighbors)), max_layer))
