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


def test_rccl_collectives_are_capturable_in_a_hipgraph(dev, monkeypatch):
    """The TP decode graph holds RCCL all-reduces / all-gathers.  A 1-GPU box cannot run tp > 1, but a
    one-rank communicator with the world-1 shortcut disabled sends the same calls through RCCL's enqueue path:
    capture them on a side stream in the capture mode the engine uses under TP, replay, check the results."""
    from mini_sglang_amd import kernel

    monkeypatch.setenv("MSGL_COMM_NO_SHORTCUT", "1")
    comm = kernel.RcclCommunicator(0, 1, 0, kernel.create_unique_id())
    try:
        x = torch.arange(4096, dtype=torch.float32, device=dev).to(torch.bfloat16)
        src = torch.full((1024,), 5, dtype=torch.bfloat16, device=dev)
        dst = torch.zeros((1024,), dtype=torch.bfloat16, device=dev)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            comm.all_reduce(x, "sum")  # warm-up outside capture (channel setup), as GraphRunner does
            comm.all_gather(dst, src)
        stream.synchronize()
        expect = x.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            y = x * 2
            comm.all_reduce(y, "sum")
            comm.all_gather(dst, src)
            z = y + 1
        dst.zero_()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(z.cpu(), (expect * 2 + 1).cpu())  # one rank: SUM all-reduce is the identity
        assert torch.equal(dst.cpu(), src.cpu())
    finally:
        comm.destroy()
