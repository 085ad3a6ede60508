"""RCCL communicator wrapper on one GPU (world_size 1): the known answers of
tests/kernel/test_comm.py:96-149 degenerate to identities, which still exercises unique-id
creation, ncclCommInitRank, dtype mapping, stream plumbing and teardown."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_comm_world1_known_answers(dev):
    from mini_sglang_amd import kernel

    uid = kernel.create_unique_id()
    assert len(uid) == 128
    comm = kernel.RcclCommunicator(0, 1, 1 << 20, uid)
    try:
        for dtype in (torch.float16, torch.bfloat16):
            x = torch.ones(8192 * 16, dtype=dtype, device=dev)
            for _ in range(4):
                comm.all_reduce(x, "sum")
            assert torch.equal(x.cpu(), torch.ones(8192 * 16, dtype=dtype))  # tp^N = 1
            src = torch.full((512,), 3, dtype=dtype, device=dev)
            dst = torch.empty((512,), dtype=dtype, device=dev)
            comm.all_gather(dst, src)
            torch.cuda.synchronize()
            assert torch.equal(dst.cpu(), src.cpu())
        with pytest.raises(ValueError):
            comm.all_reduce(x, "max")
    finally:
        comm.destroy()
