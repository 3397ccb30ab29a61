"""Transports on the GPU box: the RCCL communicator is created inside a process that already carries PyTorch's
copy of librccl (the situation of bench.py at N > 1) and real traffic goes through it -- all-reduce and a
grouped send/recv -- on ONE rank (RCCL refuses two ranks on one device; a one-GPU box cannot do more).  The
multi-rank plans themselves are covered by test_gpu_multirank.py over the shared-memory transport."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_selftest():
    import torch  # noqa: F401  (loads torch's librccl first, as bench.py does)
    from somar_amd import api

    uid = api.comm_unique_id()
    assert len(uid) == api.COMM_ID_BYTES and any(uid)
    h = api.comm_create(uid, 0, 1, 0)
    try:
        api.comm_selftest(h)
        api.comm_selftest(h)  # communicators are reused across many exchanges
    finally:
        api.comm_destroy(h)


def test_rccl_next_to_torch_process_group():
    """bench.py keeps a torch.distributed NCCL group (control plane) next to the library's communicator."""
    import torch
    import torch.distributed as dist
    from somar_amd import api

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        t = torch.ones(4, device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        h = api.comm_create(api.comm_unique_id(), 0, 1, 0)
        try:
            api.comm_selftest(h)
            dist.all_reduce(t)
            torch.cuda.synchronize()
            api.comm_selftest(h)
        finally:
            api.comm_destroy(h)
    finally:
        dist.destroy_process_group()


def test_shm_single_rank_selftest():
    from somar_amd import api

    h = api.comm_create_shm("/somar_selftest_%d" % os.getpid(), 0, 1, 1 << 20)
    try:
        api.comm_selftest(h)
    finally:
        api.comm_destroy(h)
