"""Next row 8f-4: two-hand mesh self-collision.  The reference's BVH is un-vendored (parity unpinned), so the oracle is
pinned by known-answer cases and by an independent algorithm; the GPU kernel must reproduce the oracle's pair list."""
import numpy as np
import pytest
import torch

from oracle import collision_oracle as CO


def _rand_tris(n, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 3, 3))
    b = rng.normal(size=(n, 3, 3)) + rng.normal(size=(n, 1, 3)) * spread
    return a, b


def test_sat_known_answers():
    t = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float64)
    pierce = np.array([[[0.2, 0.2, -1], [0.2, 0.2, 1], [2, 2, 0.5]]], dtype=np.float64)          # edge through the interior
    far = t + np.array([0, 0, 5.0])
    coplanar_overlap = np.array([[[0.1, 0.1, 0], [0.9, 0.1, 0], [0.1, 0.9, 0]]], dtype=np.float64)
    coplanar_apart = t + np.array([3.0, 0, 0])
    touch = np.array([[[1, 0, 0], [2, 0, 1], [2, 0, -1]]], dtype=np.float64)                       # shares only the point (1,0,0)
    parallel_above = t + np.array([0, 0, 1e-3])
    assert CO.sat_intersect(t, pierce)[0]
    assert not CO.sat_intersect(t, far)[0]
    assert CO.sat_intersect(t, coplanar_overlap)[0]
    assert not CO.sat_intersect(t, coplanar_apart)[0]
    assert CO.sat_intersect(t, touch)[0]                     # touching counts as intersecting
    assert not CO.sat_intersect(t, parallel_above)[0]


def test_sat_matches_independent_edge_pierce_test():
    """For triangles in general position, intersecting <=> an edge of one pierces the other (Moller-Trumbore): a different
    algorithm from the separating-axis test of the oracle and the kernel."""
    for seed, spread in ((0, 0.5), (1, 1.0), (2, 2.0)):
        a, b = _rand_tris(20000, seed, spread)
        sat = CO.sat_intersect(a, b)
        ep = CO.edge_pierce_intersect(a, b)
        assert 0.02 < sat.mean() < 0.9
        assert np.array_equal(sat, ep), int((sat != ep).sum())


def test_icosphere_pair_counts():
    v, f = CO.icosphere(2)                                   # 162 vertices, 320 faces
    assert v.shape == (162, 3) and f.shape == (320, 3)
    vl = (v * 0.040).astype(np.float32)
    for dx, expect_hit in ((0.200, False), (0.050, True), (0.081, False)):
        vr = (v * 0.040 + np.array([dx, 0.003, 0.001])).astype(np.float32)
        score, n = CO.non_collision_score(vl, vr, f, f)
        assert (n > 0) == expect_hit
        assert score == 100 - round(n / 640 * 100, 2)
    # a closed sphere never collides with itself: every pair found is left-right
    verts, faces = CO.build_triangles(vl, (v * 0.040 + np.array([0.05, 0.003, 0.001])).astype(np.float32), f, f)
    pairs = CO.collision_pairs(verts, faces)
    assert pairs.shape[0] > 0 and (pairs[:, 0] < 320).all() and (pairs[:, 1] >= 320).all()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [2, 3])
def test_gpu_pairs_match_oracle(level):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.collision import compute_non_collision_score, mesh_collisions
    v, f = CO.icosphere(level)
    rng = np.random.default_rng(5)
    B = 4
    vl = np.stack([(v * 0.040 * (1 + 0.1 * rng.normal(size=(1, 3)))).astype(np.float32) for _ in range(B)])
    off = np.array([[0.2, 0, 0], [0.05, 0.003, 0.001], [0.03, 0.02, -0.01], [0.0795, 0.001, 0.002]])
    vr = np.stack([(v * 0.040 + off[b]).astype(np.float32) for b in range(B)])
    counts, pairs = mesh_collisions(torch.from_numpy(vl).cuda(), torch.from_numpy(vr).cuda(), f, f, max_pairs=8192)
    counts, pairs = counts.cpu().numpy(), pairs.cpu().numpy()
    for b in range(B):
        verts, faces = CO.build_triangles(vl[b], vr[b], f, f)
        ref = CO.collision_pairs(verts, faces)
        assert counts[b] == ref.shape[0], (b, counts[b], ref.shape[0])
        assert np.array_equal(pairs[b, :counts[b]], ref)
    scores, _ = compute_non_collision_score(torch.from_numpy(vl).cuda(), f, torch.from_numpy(vr).cuda(), f)
    assert scores == [CO.non_collision_score(vl[b], vr[b], f, f)[0] for b in range(B)]
    assert counts[0] == 0 and counts[1] > 0


@pytest.mark.gpu
def test_gpu_mano_sized_meshes_and_truncation():
    """778 vertices / 1538 faces per hand (the caps): random triangle soups -- many intersections, degenerate faces -- against
    the oracle, and a pair list shorter than the count."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd import synth
    from ev2hands_amd.collision import mesh_collisions
    rng = np.random.default_rng(11)
    nv, nf = 778, 1538
    fl = synth.synth_mano_assets("left", 3)["faces"].astype(np.int64)
    fr = synth.synth_mano_assets("right", 3)["faces"].astype(np.int64)
    assert fl.shape == (nf, 3)
    vl = (rng.normal(size=(2, nv, 3)) * 0.05).astype(np.float32)
    vr = (rng.normal(size=(2, nv, 3)) * 0.05 + 0.02).astype(np.float32)
    counts, pairs = mesh_collisions(torch.from_numpy(vl).cuda(), torch.from_numpy(vr).cuda(), fl, fr, max_pairs=1000)
    counts, pairs = counts.cpu().numpy(), pairs.cpu().numpy()
    for b in range(2):
        verts, faces = CO.build_triangles(vl[b], vr[b], fl, fr)
        ref = CO.collision_pairs(verts, faces)
        assert counts[b] == ref.shape[0] and counts[b] > 1000
        assert np.array_equal(pairs[b], ref[:1000])
