"""Voxel <-> patch-token index maps for the multi-segment calorimeter geometries.

The reference patches irregular calorimeters segment by segment: ``torch.split`` the flat voxel vector by ``list_edges``, reshape each
piece to its ``(L, A, R)`` grid, ``rearrange "b c (l p1) (a p2) (r p3) -> b (l a r) (p1 p2 p3 c)"`` and concatenate the tokens
(experiments/calochallenge/calochallenge_cfm/model.py:143-173, experiments/calogan/model.py:55-84,
experiments/calohadronic/model.py:55-84).  That whole pipeline is one fixed permutation of voxel indices; this module computes it
once on the host as an int32 table ``map[n, f]`` = voxel index of feature ``f`` of token ``n``, which the HIP gather / scatter
kernels (and the fused from_patches epilogue of the last Linear) consume.  Index arithmetic only - no tensor data passes through here.
"""

from __future__ import annotations

import math

import numpy as np


def segment_patch_map(list_shape, list_edges, list_patch_shape):
    """-> (map int32 (T, P), num_patches_per_dim, num_patches_per_layer, voxels_per_sample)."""
    list_shape = [[int(v) for v in s] for s in list_shape]
    list_edges = [int(e) for e in list_edges]
    list_patch_shape = [[int(v) for v in p] for p in list_patch_shape]
    if not (len(list_shape) == len(list_edges) == len(list_patch_shape)):
        raise ValueError("list_shape, list_edges and list_patch_shape must have the same length")
    per_dim, per_layer, maps, offset = [], [], [], 0
    for i, (shape, edge, patch) in enumerate(zip(list_shape, list_edges, list_patch_shape)):
        if len(shape) != 3 or len(patch) != 3:
            raise ValueError("every segment needs a 3-D shape and patch shape")
        for L, m in zip(shape, patch):
            assert L % m == 0, f"Input size ({L}) should be divisible by patch size ({m}) in axis {i}."
        if math.prod(shape) != edge:
            raise ValueError(f"segment {i}: list_edges entry {edge} != prod(shape) {math.prod(shape)}")
        l, a, r = (s // p for s, p in zip(shape, patch))
        p1, p2, p3 = patch
        idx = offset + np.arange(edge, dtype=np.int64).reshape(l, p1, a, p2, r, p3)
        maps.append(idx.transpose(0, 2, 4, 1, 3, 5).reshape(l * a * r, p1 * p2 * p3))
        per_dim.append((l, a, r))
        per_layer.append(l * a * r)
        offset += edge
    P = maps[0].shape[1]
    if any(m.shape[1] != P for m in maps):
        raise ValueError(f"all segments must give the same patch_dim, got {[m.shape[1] for m in maps]}")
    return np.ascontiguousarray(np.concatenate(maps, 0).astype(np.int32)), per_dim, per_layer, offset


def multi_segment_meshgrid(num_patches):
    """pos_z, pos_y, pos_x (float32 numpy, length T) of ViT.create_meshgrid (reference nn/vit.py:137-154): z runs over the
    concatenation of all segments' l-grids, y / x restart per segment."""
    sum_l = sum(int(seg[0]) for seg in num_patches)
    zs, ys, xs, l0 = [], [], [], 0
    for l, a, r in num_patches:
        l, a, r = int(l), int(a), int(r)
        lg = (np.arange(l0, l0 + l, dtype=np.float32) / np.float32(sum_l)).astype(np.float32)
        ag = (np.arange(a, dtype=np.float32) / np.float32(a)).astype(np.float32)
        rg = (np.arange(r, dtype=np.float32) / np.float32(r)).astype(np.float32)
        z, y, x = np.meshgrid(lg, ag, rg, indexing="ij")
        zs.append(z.ravel()); ys.append(y.ravel()); xs.append(x.ravel())
        l0 += l
    return np.concatenate(zs), np.concatenate(ys), np.concatenate(xs)
