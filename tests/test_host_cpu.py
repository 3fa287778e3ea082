"""CPU: host-side bookkeeping of the sampler that needs no device."""


def test_plan_pool_is_bounded_and_keyed():
    from pigs_amd.sampler import _PlanPool
    pool = _PlanPool()
    for k in range(10):                                  # ten problem sizes, one dead plan each
        pool.give(("sizes", k), object())
    assert sum(len(v) for v in pool.free.values()) == pool.KEEP_TOTAL
    assert pool.take(("sizes", 0)) is None               # the oldest went back to the allocator
    a, b, c = object(), object(), object()
    for ws in (a, b, c):
        pool.give(("sizes", 9), ws)
    assert len(pool.free[("sizes", 9)]) == pool.KEEP     # per key
    assert sum(len(v) for v in pool.free.values()) <= pool.KEEP_TOTAL
    got = pool.take(("sizes", 9))
    assert got is not None and pool.take(("other", 9)) is None
