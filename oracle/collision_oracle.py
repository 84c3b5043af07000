"""TEST INFRASTRUCTURE ONLY (oracle): two-hand mesh self-collision count, CPU restatement in float64 NumPy.

Next row 8f-4 of SURVEY.md.  The reference calls the third-party BVH of `mesh-intersection==0.1.0` (torch-mesh-isect,
ev2hands.yml:121; NOT vendored in /root/reference, no tests, no golden vectors):
    /root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:128-160  compute_non_collision_score
    /root/reference/src/Ev2Hands/utils/__init__.py:106-124       compute_collision_percentage
    /root/reference/src/Ev2Hands/losses.py:60-102                CollisionLoss (training only)
What the call sites fix: the two hand meshes are concatenated (left faces, then right faces + 778), vertices are the
float32 predictions * 1000 (mm) widened to float64, `triangles = vertices[faces]`, the search tree returns pairs of
colliding triangle indices (-1 = none) and  score = 100 - round(n_pairs / n_triangles * 100, 2).

**Parity unpinned**: the tree's own arithmetic (candidate cap of `max_collisions` per triangle in traversal order, its
triangle-triangle test, its de-duplication) cannot be read or run here.  This restatement defines the quantity as the
number of unordered pairs of triangles that (a) share no vertex index and (b) intersect, by the separating-axis test
(two face normals, nine edge-edge cross products, six in-plane edge normals for parallel planes; touching counts as
intersecting), all pairs, no cap -- i.e. what the tree approximates.  It is pinned by known-answer cases and by an
independent edge-pierces-triangle test on random pairs (tests/test_collision.py)."""
from __future__ import annotations

import numpy as np

EPS_AXIS = 1e-20          # squared length below which a candidate axis is degenerate (mm^4)


def build_triangles(verts_left, verts_right, faces_left, faces_right, scale: float = 1000.0):
    """[778,3] float32 metres x2, faces [1538,3] x2 -> (vertices [V,3] float64 mm, faces [F,3] int64) of the concatenated
    mesh, scaled like evaluate_ev2hands_r.py:137-138 (float32 multiply, then float64)."""
    vl = (np.asarray(verts_left, dtype=np.float32) * np.float32(scale)).astype(np.float64)
    vr = (np.asarray(verts_right, dtype=np.float32) * np.float32(scale)).astype(np.float64)
    fl = np.asarray(faces_left, dtype=np.int64)
    fr = np.asarray(faces_right, dtype=np.int64) + vl.shape[0]
    return np.concatenate([vl, vr]), np.concatenate([fl, fr])


def _separated(a, b, axis):
    """a, b [n,3,3] triangles, axis [n,3]: True where the projections are strictly disjoint (degenerate axes: False)."""
    pa = np.einsum("nvk,nk->nv", a, axis)
    pb = np.einsum("nvk,nk->nv", b, axis)
    ok = np.einsum("nk,nk->n", axis, axis) >= EPS_AXIS
    return ok & ((pa.max(1) < pb.min(1)) | (pb.max(1) < pa.min(1)))


def sat_intersect(a, b):
    """Separating-axis triangle-triangle test, vectorised: a, b [n,3,3] float64 -> bool [n]."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ea = np.stack([a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 0] - a[:, 2]], 1)
    eb = np.stack([b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 0] - b[:, 2]], 1)
    na = np.cross(ea[:, 0], ea[:, 1])
    nb = np.cross(eb[:, 0], eb[:, 1])
    sep = _separated(a, b, na) | _separated(a, b, nb)
    for i in range(3):
        for j in range(3):
            sep |= _separated(a, b, np.cross(ea[:, i], eb[:, j]))
    for i in range(3):                       # parallel planes: the edge-edge axes vanish, use the in-plane edge normals
        sep |= _separated(a, b, np.cross(na, ea[:, i]))
        sep |= _separated(a, b, np.cross(nb, eb[:, i]))
    return ~sep


def collision_pairs(vertices, faces, max_per_triangle: int = 0):
    """All unordered pairs (i < j, lexicographic order) of triangles without a common vertex index that intersect.
    max_per_triangle > 0 keeps at most that many pairs per triangle i (the first ones in j order): the role of the reference
    BVH's `max_collisions` cap (which pairs ITS traversal keeps beyond the cap is not reproducible: parity unpinned)."""
    tri = vertices[faces]                                     # [F,3,3]
    lo, hi = tri.min(1), tri.max(1)
    F = faces.shape[0]
    out = []
    for i in range(F - 1):
        j = np.arange(i + 1, F)
        ov = np.all((lo[i] <= hi[j]) & (lo[j] <= hi[i]), axis=1)
        j = j[ov]
        if j.size == 0:
            continue
        share = (faces[j][:, :, None] == faces[i][None, None, :]).any((1, 2))
        j = j[~share]
        if j.size == 0:
            continue
        hit = sat_intersect(np.broadcast_to(tri[i], (j.size, 3, 3)), tri[j])
        js = j[hit]
        if max_per_triangle > 0:
            js = js[:max_per_triangle]
        out += [(i, int(k)) for k in js]
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def non_collision_score(verts_left, verts_right, faces_left, faces_right, max_collisions: int = 0):
    """evaluate_ev2hands_r.py:149-157 with the pair count defined above (max_collisions: the BVH cap, 8 there)."""
    v, f = build_triangles(verts_left, verts_right, faces_left, faces_right)
    n = collision_pairs(v, f, max_collisions).shape[0]
    return 100 - round(n / f.shape[0] * 100, 2), n


# ---- penetration penalty of the training loss (losses.py:60-102)
def cone_term(face, pts, sigma: float = 0.5):
    """Conic distance-field term of one triangle `face` [3,3] for the points `pts` [n,3] (float64): Tzionas et al., IJCV 2016,
    eq. 11-14, the definition torch-mesh-isect's DistanceFieldPenetrationLoss implements (sigma = 0.5, point2plane = False,
    penalize_outside = False at the reference's call site, losses.py:70-72; the package is not vendored: PARITY UNPINNED).
    o, r = circumcentre / circumradius of the face, n its unit normal; a point at depth h = -n.(p - o) >= 0 behind the face is
    inside the cone when Phi = |radial offset| / (r (1 + h / sigma)) < 1 and then costs Psi^2 with Psi = (1 - Phi)^2; points in
    front of the face (h < 0) or outside the cone cost nothing.  Returns the sum over the points."""
    f = np.asarray(face, dtype=np.float64)
    a, b = f[1] - f[0], f[2] - f[0]
    axb = np.cross(a, b)
    n2 = axb @ axb
    if n2 < 1e-300:
        return 0.0
    o = f[0] + np.cross((a @ a) * b - (b @ b) * a, axb) / (2 * n2)
    r = np.linalg.norm(a) * np.linalg.norm(b) * np.linalg.norm(a - b) / (2 * np.sqrt(n2))
    n = axb / np.sqrt(n2)
    d = np.asarray(pts, dtype=np.float64) - o
    along = d @ n
    rad = np.linalg.norm(d - along[:, None] * n, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):      # the cone's apex (along = sigma) lies in front of the face: masked below
        phi = rad / (r * (1.0 - along / sigma))
    psi = np.where((along <= 0) & (phi < 1), (1 - phi) ** 2, 0.0)
    return float((psi ** 2).sum())


def penetration_loss(vertices, faces, pairs, sigma: float = 0.5):
    """Sum over the colliding pairs of both triangles' cone terms evaluated at the other triangle's vertices (one window)."""
    tri = np.asarray(vertices, dtype=np.float64)[faces]
    return float(sum(cone_term(tri[i], tri[j], sigma) + cone_term(tri[j], tri[i], sigma) for i, j in pairs))


def collision_loss(verts_left, verts_right, faces_left, faces_right, max_collisions: int = 16, sigma: float = 0.5, weight: float = 1e2,
                   reference_batch_quirk: bool = False):
    """CollisionLoss.__call__ (losses.py:77-102) for a batch: verts_* [B,778,3] float32 METRES (the loss does not rescale).
    Returns (loss value, per-window penalties).
    Every window uses its OWN vertices.  Upstream does not for B > 1: losses.py:88-93 indexes `verts_tensor.view([-1, 3])` with
    face indices that are not offset per batch item, so all B triangle sets are built from item 0's vertices and the upstream
    value is 100 x window 0's penalty; reference_batch_quirk=True restates that."""
    if reference_batch_quirk:
        verts_left = [verts_left[0]] * len(verts_left)
        verts_right = [verts_right[0]] * len(verts_right)
    per = []
    for vl, vr in zip(verts_left, verts_right):
        v, f = build_triangles(vl, vr, faces_left, faces_right, scale=1.0)
        per.append(penetration_loss(v, f, collision_pairs(v, f, max_collisions), sigma))
    per = np.asarray(per)
    nz = per[per != 0]
    return (float(nz.mean() * weight) if nz.size else 0.0), per


# ---- independent check used by the tests: two triangles in general position intersect iff an edge of one pierces the other
def _segment_hits_triangle(p, q, t):
    """Moller-Trumbore on the segment p->q against triangle t; p, q [n,3], t [n,3,3] -> bool [n] (proper crossings)."""
    d = q - p
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    h = np.cross(d, e2)
    det = np.einsum("nk,nk->n", e1, h)
    ok = np.abs(det) > 1e-12
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    s = p - t[:, 0]
    u = np.einsum("nk,nk->n", s, h) * inv
    qv = np.cross(s, e1)
    v = np.einsum("nk,nk->n", d, qv) * inv
    w = np.einsum("nk,nk->n", e2, qv) * inv
    return ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (w >= 0) & (w <= 1)


def edge_pierce_intersect(a, b):
    hit = np.zeros(a.shape[0], dtype=bool)
    for i in range(3):
        hit |= _segment_hits_triangle(a[:, i], a[:, (i + 1) % 3], b)
        hit |= _segment_hits_triangle(b[:, i], b[:, (i + 1) % 3], a)
    return hit


def icosphere(level: int = 3):
    """Unit icosphere: level 3 -> 642 vertices, 1280 faces (fits the kernel's 778 / 1538 caps)."""
    t = (1.0 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(x, dtype=np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(i, j):
            key = (min(i, j), max(i, j))
            if key not in cache:
                m = v[i] + v[j]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v), np.asarray(f, dtype=np.int64)
