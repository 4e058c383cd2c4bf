"""Objects produced on one context and consumed on another: in stream-ordered mode an index build / attribute estimation
returns while its kernels are still in flight; a consumer on ANOTHER context must wait for the producer's event (common.hpp: Ready), so
the result cannot depend on which context consumes what."""
import numpy as np
import pytest

import small_gicp_amd as sga

pytestmark = pytest.mark.gpu


def test_stream_ordered_producer_and_a_consumer_on_another_context():
    target, source, T_gt = sga.synthetic.registration_pair(200_000)
    st = sga.make_setting("GICP", max_correspondence_distance=1.0)
    # reference: everything on one synchronising context
    c0 = sga.Context(0)
    tgt0, src0 = sga.PointCloud(target, ctx=c0), sga.PointCloud(source, ctx=c0)
    sga.estimate_covariances(tgt0, None, 20)
    sga.estimate_covariances(src0, None, 20)
    want = sga.Problem(sga.KdTree(tgt0), src0, ctx=c0).align(st, np.eye(4))
    # producer in stream-ordered mode, consumer on a second context, no synchronisation by the caller in between
    for rep in range(3):
        prod, cons = sga.Context(0), sga.Context(0)
        prev = prod.set_stream_ordered(True)
        assert prev is False
        tgt, src = sga.PointCloud(target, ctx=prod), sga.PointCloud(source, ctx=prod)
        tree = sga.KdTree(tgt)
        sga.estimate_covariances(tgt, tree, 20)
        stree = sga.KdTree(src)
        sga.estimate_covariances(src, stree, 20)
        got = sga.Problem(tree, stree, ctx=cons).align(st, np.eye(4))  # the source by its own index: borrows arrays that may still be in flight
        got2 = sga.Problem(tree, src, ctx=cons).align(st, np.eye(4))
        prod.set_stream_ordered(False)
        for g in (got, got2):
            assert g.iterations == want.iterations and g.num_inliers == want.num_inliers
            assert np.abs(g.T_target_source - want.T_target_source).max() < 1e-6


def test_online_odometry_leaves_the_default_context_alone():
    from small_gicp_amd import odometry

    d = sga.default_context()
    assert getattr(d, "stream_ordered", False) is False
    od = odometry.OnlineOdometry()
    assert od.ctx is not d
    pts, _ = sga.synthetic.kitti_like_scan(0)
    od.estimate(pts)
    assert getattr(d, "stream_ordered", False) is False
    borrowed = sga.Context(0)
    od2 = odometry.OnlineOdometry(ctx=borrowed)
    assert borrowed.stream_ordered is True
    od2.close()
    assert borrowed.stream_ordered is False


def test_registrations_side_by_side_share_an_index_and_agree():
    """Three host threads, a context (stream) and a factor state each, ONE target index and ONE source cloud between them (bench.py:
    concurrent_registrations): every registration of every job must give the pose of a lone registration, bit for bit — nothing a pass
    writes (certificates, partial rows, tile orders, hand-off slots) may be shared between problems or contexts."""
    import threading

    target, source, T_gt = sga.synthetic.registration_pair(150_000)
    c0 = sga.Context(0)
    tgt, src = sga.PointCloud(target, ctx=c0), sga.PointCloud(source, ctx=c0)
    sga.estimate_covariances(tgt, None, 20)
    sga.estimate_covariances(src, None, 20)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
    lone = sga.Problem(tree, src, ctx=c0)
    lone.align(st, np.eye(4))
    want = lone.align(st, np.eye(4))  # a registration that follows another one (tile orders, warm buffers in their steady state)
    J, regs = 3, 6
    ctxs = [sga.Context(0) for _ in range(J)]
    pbs = [sga.Problem(tree, src, ctx=c) for c in ctxs]
    got, errors = [[] for _ in range(J)], []
    gate = threading.Barrier(J)

    def work(j):
        try:
            gate.wait()
            for _ in range(regs):
                got[j].append(pbs[j].align(st, np.eye(4)))
        except BaseException as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=work, args=(j,)) for j in range(J)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for j in range(J):
        assert len(got[j]) == regs
        for r in got[j][1:]:  # (the first registration of a fresh problem has no tile order yet: equal to ~1e-9, not bit for bit)
            assert r.iterations == want.iterations and r.num_inliers == want.num_inliers
            assert np.array_equal(r.T_target_source, want.T_target_source)
        assert np.abs(got[j][0].T_target_source - want.T_target_source).max() < 1e-6
