"""Host CPU budget helper (patchaugnet_amd/hostcpu.py)."""
import os

import torch

from patchaugnet_amd import hostcpu


def test_budget_within_affinity():
    n = hostcpu.cpu_budget()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_quota_parsing(monkeypatch):
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: 2.5)
    assert hostcpu.cpu_budget() == min(2, len(os.sched_getaffinity(0)))
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: 0.3)
    assert hostcpu.cpu_budget() == 1
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: None)
    assert hostcpu.cpu_budget() == len(os.sched_getaffinity(0))


def test_limit_applies():
    before = torch.get_num_threads()
    try:
        assert hostcpu.limit_host_threads(1) == 1
        assert torch.get_num_threads() == 1 and os.environ["OMP_NUM_THREADS"] == "1"
    finally:
        hostcpu.limit_host_threads(before)
