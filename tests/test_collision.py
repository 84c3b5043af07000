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
    scores, _ = compute_non_collision_score(torch.from_numpy(vl).cuda(), f, torch.from_numpy(vr).cuda(), f, max_collisions=0)
    assert scores == [CO.non_collision_score(vl[b], vr[b], f, f)[0] for b in range(B)]
    scores8, _ = compute_non_collision_score(torch.from_numpy(vl).cuda(), f, torch.from_numpy(vr).cuda(), f)       # the reference's cap of 8
    assert scores8 == [CO.non_collision_score(vl[b], vr[b], f, f, 8)[0] for b in range(B)]
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


def test_cone_term_known_answers():
    """Conic distance field (Tzionas et al. eq. 11-14): a right triangle with legs 2 has circumcentre (1, 1, 0) and
    circumradius sqrt(2)."""
    f = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0]], dtype=np.float64)           # normal +z
    sig = 0.5
    assert CO.cone_term(f, np.array([[1.0, 1.0, 0.0]]), sig) == pytest.approx(1.0)                 # on the axis, on the face: Psi = 1
    assert CO.cone_term(f, np.array([[1.0, 1.0, 0.5]]), sig) == 0.0                                # in front of the face: free
    assert CO.cone_term(f, np.array([[1.0 + np.sqrt(2), 1.0, 0.0]]), sig) == pytest.approx(0.0)    # on the circumcircle: Phi = 1
    # depth h = 0.5 = sigma doubles the cone radius: a point at radial distance sqrt(2) has Phi = 1/2, Psi = 1/4, cost 1/16
    assert CO.cone_term(f, np.array([[1.0 + np.sqrt(2), 1.0, -0.5]]), sig) == pytest.approx(1.0 / 16)
    # rigid motion invariance and additivity over points
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(5, 3)) * 0.5 + np.array([1, 1, -0.3])
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    R *= np.sign(np.linalg.det(R))
    t = rng.normal(size=3)
    assert CO.cone_term(f @ R.T + t, pts @ R.T + t, sig) == pytest.approx(CO.cone_term(f, pts, sig), rel=1e-12)
    assert CO.cone_term(f, pts, sig) == pytest.approx(sum(CO.cone_term(f, p[None], sig) for p in pts), rel=1e-12)


def test_collision_loss_reduction_matches_the_call_site():
    """losses.py:96-100: mean over the windows with a non-zero penalty, times 100; 0 when nothing collides."""
    v, f = CO.icosphere(2)
    vl = np.stack([(v * 0.040).astype(np.float32)] * 3)
    vr = np.stack([(v * 0.040 + np.array(o)).astype(np.float32) for o in ([0.2, 0, 0], [0.05, 0.003, 0.001], [0.03, 0.02, -0.01])])
    loss, per = CO.collision_loss(vl, vr, f, f)
    assert per[0] == 0 and per[1] > 0 and per[2] > 0
    assert loss == pytest.approx(per[1:].mean() * 100)
    assert CO.collision_loss(vl[:1], vr[:1], f, f)[0] == 0.0


@pytest.mark.gpu
def test_gpu_collision_loss_matches_oracle():
    """ev2h_mesh_collisions (scale 1, cap 16) + ev2h_collision_penalty + the call site's reduction against the float64 oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.collision import CollisionLoss, mesh_collisions
    v, f = CO.icosphere(3)
    rng = np.random.default_rng(7)
    B = 5
    vl = np.stack([(v * 0.040 * (1 + 0.05 * rng.normal(size=(1, 3)))).astype(np.float32) for _ in range(B)])
    off = np.array([[0.2, 0, 0], [0.05, 0.003, 0.001], [0.03, 0.02, -0.01], [0.0795, 0.001, 0.002], [0.01, 0.0, 0.0]])
    vr = np.stack([(v * 0.040 + off[b]).astype(np.float32) for b in range(B)])
    outs = {"left": {"vertices": torch.from_numpy(vl).cuda(), "faces": f}, "right": {"vertices": torch.from_numpy(vr).cuda(), "faces": f}}
    cl = CollisionLoss("cuda:0")
    per = cl.per_window(outs).cpu().numpy()
    want, want_per = CO.collision_loss(vl, vr, f, f)
    assert want_per[0] == 0 and (want_per[1:3] > 0).all()
    assert np.allclose(per, want_per, rtol=1e-9, atol=1e-300)
    assert float(cl(outs)) == pytest.approx(want, rel=1e-6)
    # the cap of 16 pairs per triangle is active for the deeply interpenetrating window
    c16, _ = mesh_collisions(outs["left"]["vertices"], outs["right"]["vertices"], f, f, scale=1.0, max_per_triangle=16)
    c0, _ = mesh_collisions(outs["left"]["vertices"], outs["right"]["vertices"], f, f, scale=1.0)
    assert (c16 <= c0).all() and int(c16[4]) <= int(c0[4])
    for b in (1, 4):
        vv, ff = CO.build_triangles(vl[b], vr[b], f, f, scale=1.0)
        assert int(c16[b]) == CO.collision_pairs(vv, ff, 16).shape[0]


def test_reference_batch_quirk_is_restated():
    """losses.py:88-93 indexes the flattened vertices of the whole batch with un-offset face indices: upstream's value for B > 1 is
    100 x window 0's penalty.  Default here: every window its own vertices; the flag restates upstream."""
    v, f = CO.icosphere(2)
    vl = np.stack([(v * 0.040).astype(np.float32)] * 3)
    vr = np.stack([(v * 0.040 + o).astype(np.float32) for o in ([0.05, 0.003, 0.001], [0.2, 0, 0], [0.03, 0.02, -0.01])])
    own, per = CO.collision_loss(vl, vr, f, f)
    quirk, perq = CO.collision_loss(vl, vr, f, f, reference_batch_quirk=True)
    assert per[0] > 0 and per[1] == 0 and per[2] > 0
    assert np.all(perq == per[0]) and quirk == pytest.approx(per[0] * 100) and own != pytest.approx(quirk)


@pytest.mark.gpu
def test_gpu_collision_loss_capacity_and_quirk():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.collision import CollisionLoss, device_faces
    v, f = CO.icosphere(3)
    B = 3
    vl = np.stack([(v * 0.040).astype(np.float32)] * B)
    vr = np.stack([(v * 0.040 + o).astype(np.float32) for o in ([0.01, 0.0, 0.0], [0.2, 0, 0], [0.03, 0.02, -0.01])])
    outs = {"left": {"vertices": torch.from_numpy(vl).cuda(), "faces": f}, "right": {"vertices": torch.from_numpy(vr).cuda(), "faces": f}}
    cl = CollisionLoss("cuda:0")
    assert cl.capacity(f.shape[0]) == 2 * f.shape[0] * 16
    per = cl.per_window(outs).cpu().numpy()
    assert not cl.truncated()
    want, want_per = CO.collision_loss(vl, vr, f, f)
    assert np.allclose(per, want_per, rtol=1e-9, atol=1e-300)
    # device face tensors prepared once give the same result and are passed through without a copy
    df = device_faces(f, "cuda:0")
    assert np.array_equal(cl.per_window(outs, faces=(df, df)).cpu().numpy(), per)
    # a deliberately small pair list is reported as truncated
    small = CollisionLoss("cuda:0", max_pairs=4)
    small.per_window(outs)
    assert small.truncated()
    q = CollisionLoss("cuda:0", reference_batch_quirk=True)
    wantq, _ = CO.collision_loss(vl, vr, f, f, reference_batch_quirk=True)
    assert float(q(outs)) == pytest.approx(wantq, rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [1, 3, 16])
def test_gpu_capped_pair_list_one_walk_equals_two_walks_and_the_oracle(cap):
    """With a per-triangle cap and a list of >= n_triangles * cap entries the kernel writes each row's pairs into a slot range
    during the counting walk and compacts in place (one walk); with a smaller list it counts, prefix-sums and walks again.  Both
    must give the oracle's capped list, entry for entry, also when the cap is active on many rows."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.collision import mesh_collisions
    v, f = CO.icosphere(3)
    B = 4
    rng = np.random.default_rng(3)
    vl = np.stack([(v * 0.040 * (1 + 0.05 * rng.normal(size=(1, 3)))).astype(np.float32) for _ in range(B)])
    off = np.array([[0.2, 0, 0], [0.05, 0.003, 0.001], [0.01, 0.0, 0.0], [0.0, 0.0005, 0.0]])
    vr = np.stack([(v * 0.040 + off[b]).astype(np.float32) for b in range(B)])
    tl, tr = torch.from_numpy(vl).cuda(), torch.from_numpy(vr).cuda()
    F2 = 2 * f.shape[0]
    c1, p1 = mesh_collisions(tl, tr, f, f, max_pairs=F2 * cap, scale=1.0, max_per_triangle=cap)            # one walk
    c2, p2 = mesh_collisions(tl, tr, f, f, max_pairs=F2 * cap - 1, scale=1.0, max_per_triangle=cap)        # two walks
    assert torch.equal(c1, c2)
    for b in range(B):
        vv, ff = CO.build_triangles(vl[b], vr[b], f, f, scale=1.0)
        ref = CO.collision_pairs(vv, ff, cap)
        n = int(c1[b])
        assert n == ref.shape[0]
        assert np.array_equal(p1[b, :n].cpu().numpy(), ref)
        m = min(n, F2 * cap - 1)
        assert np.array_equal(p2[b, :m].cpu().numpy(), ref[:m])
    assert int(c1[0]) == 0 and int(c1[2]) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("mesh", ["surface", "soup"])
def test_gpu_two_workgroups_per_window_equal_one(mesh):
    """ev2h_mesh_collisions_ws with a scratch buffer splits a window's row blocks over two workgroups (B <= 128: BASELINE config 5
    runs 128 windows per GPU on 256 CUs); without one it is the single-workgroup search.  Counts and pair lists must be identical
    entry for entry in all three forms of the call: count only, capped one-walk list, uncapped / short two-walk list."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from ev2hands_amd import _lib, synth
    from ev2hands_amd.mano import ManoHand
    B, nv, nf = 6, 778, 1538
    make = synth.synth_mano_surface_assets if mesh == "surface" else synth.synth_mano_assets
    hands = {s: ManoHand(make(s, 2), "cuda") for s in ("left", "right")}
    g = lambda n, sc: torch.from_numpy(synth.hash_normal("c" + str(n), (B, n), 7) * sc).float().cuda()      # noqa: E731
    out = {s: hands[s](g(3, 0.4), g(6, 0.6), g(10, 0.5), g(3, 0.01)) for s in hands}
    vl, vr = out["left"].vertices.contiguous(), (out["right"].vertices + torch.tensor([0.03, 0.0, 0.0], device="cuda")).contiguous()
    fl = torch.as_tensor(np.asarray(hands["left"].faces).astype(np.int32)).cuda()
    fr = torch.as_tensor(np.asarray(hands["right"].faces).astype(np.int32)).cuda()
    L = _lib.lib()
    scratch = torch.empty(L.ev2h_mesh_collisions_scratch_bytes(B, nf), dtype=torch.uint8, device="cuda")

    def run(max_pairs, cap, ws):
        counts = torch.full((B,), -1, device="cuda", dtype=torch.int32)
        pairs = torch.full((B, max(max_pairs, 1), 2), -1, device="cuda", dtype=torch.int32)
        _lib.check(L.ev2h_mesh_collisions_ws(vl.data_ptr(), vr.data_ptr(), fl.data_ptr(), fr.data_ptr(), B, nv, nf, 1000.0, max_pairs,
                                             pairs.data_ptr() if max_pairs else None, counts.data_ptr(), cap,
                                             scratch.data_ptr() if ws else None, scratch.numel() if ws else 0, _lib.stream_handle()), "ws")
        torch.cuda.synchronize()
        return counts.cpu(), pairs.cpu()

    for max_pairs, cap in ((0, 8), (2 * nf * 16, 16), (5000, 0), (2 * nf * 2 - 1, 2)):
        c1, p1 = run(max_pairs, cap, False)
        c2, p2 = run(max_pairs, cap, True)
        assert torch.equal(c1, c2), (max_pairs, cap, c1, c2)
        for b in range(B):
            n = min(int(c1[b]), max_pairs)
            assert torch.equal(p1[b, :n], p2[b, :n]), (max_pairs, cap, b)
    print(f"{mesh}: pairs per window {c1.tolist()}")
    assert int(c1.max()) > 0


@pytest.mark.gpu
def test_gpu_searches_on_two_streams_do_not_share_scratch():
    """mesh_collisions' cached scratch is per (device, stream): two searches in flight on different streams used to share one
    buffer (row counters of one search overwritten by the other's phases -- ADVICE r4).  Many interleaved searches on two streams
    must reproduce the single-stream lists, and a caller-owned scratch buffer is accepted."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd import collision as col, synth
    from ev2hands_amd.mano import ManoHand
    B = 5
    hands = {s: ManoHand(synth.synth_mano_surface_assets(s, 2), "cuda") for s in ("left", "right")}
    g = lambda n, sc, sd: torch.from_numpy(synth.hash_normal("c" + str(n), (B, n), sd) * sc).float().cuda()      # noqa: E731
    sets = []
    for sd in (7, 8):
        out = {s: hands[s](g(3, 0.4, sd), g(6, 0.6, sd), g(10, 0.5, sd), g(3, 0.01, sd)) for s in hands}
        sets.append((out["left"].vertices.contiguous(), (out["right"].vertices + torch.tensor([0.03, 0.0, 0.0], device="cuda")).contiguous()))
    fl, fr = hands["left"].faces, hands["right"].faces
    cap, mp = 16, 2 * 1538 * 16
    ref = [col.mesh_collisions(vl, vr, fl, fr, max_pairs=mp, scale=1.0, max_per_triangle=cap) for vl, vr in sets]
    torch.cuda.synchronize()
    ref = [(c.cpu(), p.cpu()) for c, p in ref]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    got = [[], []]
    for _ in range(6):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                got[i].append(col.mesh_collisions(*sets[i], fl, fr, max_pairs=mp, scale=1.0, max_per_triangle=cap))
    torch.cuda.synchronize()
    assert len({k for k in col._SCRATCH if k[0].startswith("cuda")}) >= 3          # default stream + the two side streams
    for i in range(2):
        for c, p in got[i]:
            assert torch.equal(c.cpu(), ref[i][0])
            for b in range(B):
                n = int(ref[i][0][b])
                assert torch.equal(p[b, :n].cpu(), ref[i][1][b, :n])
    own = torch.empty(_scratch_bytes(B), dtype=torch.uint8, device="cuda")
    c, p = col.mesh_collisions(*sets[0], fl, fr, max_pairs=mp, scale=1.0, max_per_triangle=cap, scratch=own)
    assert torch.equal(c.cpu(), ref[0][0])
    with pytest.raises(ValueError):
        col.mesh_collisions(*sets[0], fl, fr, scratch=torch.empty(16, dtype=torch.float32, device="cuda"))


def _scratch_bytes(B, nf=1538):
    from ev2hands_amd import _lib
    return _lib.lib().ev2h_mesh_collisions_scratch_bytes(B, nf)
