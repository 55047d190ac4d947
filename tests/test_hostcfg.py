"""Host thread budget (robosumo_selfplay_amd/hostcfg.py): the cgroup CPU quota is what the pools must be sized by, not os.cpu_count()
(the cause of the bimodal short bench window of round 1: DESIGN.md §4, bench reproducibility)."""
import builtins
import io
import os

from robosumo_selfplay_amd import hostcfg


def _fake_open(files):
    real = builtins.open

    def f(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        return real(path, *a, **k)
    return f


def test_cpu_quota_reads_cgroup_v2_and_v1(monkeypatch):
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert hostcfg.cpu_quota() == 16
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert hostcfg.cpu_quota() == 256
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
                                                       "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert hostcfg.cpu_quota() == 8
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "50000 100000\n"}))
    assert hostcfg.cpu_quota() == 1                                      # never below one
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(4)), raising=False)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert hostcfg.cpu_quota() == 4                                      # the affinity mask caps it too


def test_apply_caps_pools_and_respects_overrides(monkeypatch):
    for v in hostcfg._VARS:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.delenv("SUMO_HOST_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(hostcfg, "_applied", None)
    monkeypatch.setattr(hostcfg, "cpu_quota", lambda: 16)
    assert hostcfg.apply() == 8 and os.environ["OMP_NUM_THREADS"] == "8" and os.environ["OPENBLAS_NUM_THREADS"] == "8"
    assert hostcfg.apply() == 8                                           # idempotent
    monkeypatch.setattr(hostcfg, "_applied", None)
    for v in hostcfg._VARS:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")                           # four ranks share the node's quota
    assert hostcfg.apply() == 2 and os.environ["OPENBLAS_NUM_THREADS"] == "2"
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "64")
    monkeypatch.setattr(hostcfg, "_applied", None)
    assert hostcfg.apply() == 1                                           # never below one
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(hostcfg, "_applied", None)
    monkeypatch.setenv("SUMO_HOST_THREADS", "0")
    assert hostcfg.apply() == 0                                           # opt-out
    monkeypatch.setattr(hostcfg, "_applied", None)
    monkeypatch.setenv("SUMO_HOST_THREADS", "3")
    monkeypatch.setenv("OMP_NUM_THREADS", "5")                            # a user's own setting is not overridden
    assert hostcfg.apply() == 3 and os.environ["OMP_NUM_THREADS"] == "5"


def test_throttle_stats_parses_cpu_stat(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.stat": "usage_usec 10\nnr_periods 5\nnr_throttled 3\nthrottled_usec 12345\n"}))
    assert hostcfg.throttle_stats() == (3, 12345)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.stat": None}))
    assert hostcfg.throttle_stats() is None


def test_gc_paused_restores_collector_state():
    import gc
    from robosumo_selfplay_amd import hostcfg
    assert gc.isenabled()
    with hostcfg.gc_paused():
        assert not gc.isenabled()
        with hostcfg.gc_paused():                 # nested (a capture inside a paused region) keeps it off
            assert not gc.isenabled()
        assert not gc.isenabled()
    assert gc.isenabled()
    gc.disable()
    try:
        with hostcfg.gc_paused():
            pass
        assert not gc.isenabled()                 # was off before: stays off
    finally:
        gc.enable()


def test_drop_graphs_refuses_inside_a_capture_block():
    """A captured HIP graph destroyed while a stream is capturing aborts the process: the release path raises instead."""
    import pytest
    from robosumo_selfplay_amd import hostcfg
    graphs = {"k": object()}
    assert not hostcfg.capturing()
    with hostcfg.gc_paused():
        assert hostcfg.capturing()
        with pytest.raises(RuntimeError, match="graph capture is open"):
            hostcfg.drop_graphs(graphs)
        assert graphs                              # nothing was released
    assert not hostcfg.capturing()
    hostcfg.drop_graphs(graphs)
    assert not graphs
