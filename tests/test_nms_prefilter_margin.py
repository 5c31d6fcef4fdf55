"""CPU check of the margin argument in csrc/detect.hip iou_col_suppressed: Fast NMS only needs `IoU <= thresh`, so the kernel
decides with q = inter * rcp(union) (v_rcp_f32: 1 ulp) and falls back to the IEEE division only when q lies within 2^-20 of the
threshold (or the union is not a normal number).  Here: for many box pairs, every quotient the hardware could produce (the
reciprocal perturbed by -1 / 0 / +1 ulp, the product rounded to fp32) classifies exactly like the correctly rounded division
whenever it is outside that band — i.e. the pre-test can never flip a decision."""
import numpy as np

F = np.float32


def _pairs(n, rng):
    c = rng.random((n, 2, 2)).astype(F)
    wh = (rng.random((n, 2, 2)) * np.array([0.3, 0.3]) + 1e-3).astype(F)
    a, b = np.concatenate([c[:, 0], c[:, 0] + wh[:, 0]], 1), np.concatenate([c[:, 1], c[:, 1] + wh[:, 1]], 1)
    # half of the pairs: b = a slightly shifted / scaled, so that IoUs crowd around typical thresholds
    k = n // 2
    jitter = (rng.standard_normal((k, 4)) * 0.03).astype(F)
    b[:k] = a[:k] + jitter * (a[:k, 2:4] - a[:k, 0:2]).repeat(2, 1).reshape(k, 4)[:, [0, 1, 0, 1]]
    return a.astype(F), b.astype(F)


def test_reciprocal_pretest_never_flips_a_decision():
    rng = np.random.default_rng(0)
    a, b = _pairs(400000, rng)
    iw = np.maximum(np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]), F(0)).astype(F)
    ih = np.maximum(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]), F(0)).astype(F)
    inter = (iw * ih).astype(F)
    area_a = ((a[:, 2] - a[:, 0]).astype(F) * (a[:, 3] - a[:, 1]).astype(F)).astype(F)
    area_b = ((b[:, 2] - b[:, 0]).astype(F) * (b[:, 3] - b[:, 1]).astype(F)).astype(F)
    uni = ((area_a + area_b).astype(F) - inter).astype(F)
    ok = (uni > F(1e-30)) & (uni < F(1e30))
    exact = (inter / uni).astype(F)                                  # IEEE division, what jaccard() computes
    checked = 0
    for thresh in (F(0.5), F(0.3), F(0.45), F(0.7)):
        eps = F(abs(thresh) * 2.0 ** -20)
        lo, hi = F(thresh - eps), F(thresh + eps)
        r0 = (F(1) / uni).astype(F)
        for d in (-1, 0, 1):                                         # any reciprocal within one ulp of the rounded one
            r = np.nextafter(r0, F(np.inf) if d > 0 else F(-np.inf)).astype(F) if d else r0
            q = (inter * r).astype(F)
            below, above = ok & (q < lo), ok & (q > hi)
            assert not (exact[below] > thresh).any()                # "clearly at or below" is never a suppression
            assert (exact[above] > thresh).all()                    # "clearly above" always is
            checked += int(below.sum() + above.sum())
        band = ok & ~((inter * r0).astype(F) < lo) & ~((inter * r0).astype(F) > hi)
        assert band.sum() < 50                                       # the exact path is the rare one
    assert checked > 4e6
