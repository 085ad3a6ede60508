"""TEST INFRASTRUCTURE: record the batches an engine is handed and replay them, teacher-forced, through another engine
(same rows, lengths, input ids, positions, out_loc and page-table rows), so that logits are comparable forward by
forward no matter how sampling would have diverged."""
from __future__ import annotations

from typing import Any, Dict, List

import torch


def record_offline_runner(runner: Any, engine: Any, forwards: List[Dict[str, Any]]) -> None:
    """Wrap OfflineRunner._forward / Sampler.sample of the repo engine to record every forward."""
    orig_fwd, orig_sample = runner._forward, engine.sampler.sample

    def fwd(batch, write, args):
        padded = batch.padded_reqs
        rows = [r.table_idx for r in padded]
        max_len = max(r.device_len for r in padded)
        forwards.append(dict(
            phase=batch.phase, size=batch.size, padded_size=batch.padded_size, rows=rows, uids=[r.uid for r in padded],
            cached_lens=[r.cached_len for r in padded], device_lens=[r.device_len for r in padded],
            input_ids=batch.input_ids.cpu(), positions=batch.positions.cpu(), out_loc=batch.out_loc.cpu(),
            table=engine.page_table[torch.tensor(rows, device=engine.device)][:, :max_len].cpu(),
            graph=bool(engine.graph_runner.can_use_cuda_graph(batch))))
        return orig_fwd(batch, write, args)

    def sample(logits, args):
        forwards[-1]["logits"] = logits.float().cpu()
        return orig_sample(logits, args)

    runner._forward, engine.sampler.sample = fwd, sample


def replay_forward(eng: Any, f: Dict[str, Any]) -> torch.Tensor:
    """One recorded forward through `eng` (page-table rows restored first); returns logits[:size]."""
    from mini_sglang_amd.core import Batch, Req

    dev = eng.device
    rows = torch.tensor(f["rows"], device=dev)
    table = f["table"].to(dev)
    eng.page_table[rows, : table.shape[1]] = table
    reqs = [Req(input_ids=torch.zeros(dl, dtype=torch.int32), table_idx=row, cached_len=cl, output_len=1 << 20, uid=uid)
            for row, cl, dl, uid in zip(f["rows"], f["cached_lens"], f["device_lens"], f["uids"])]
    batch = Batch(reqs=reqs[: f["size"]], phase=f["phase"])
    batch.padded_reqs = reqs
    batch.input_ids, batch.positions, batch.out_loc = (f[k].to(dev) for k in ("input_ids", "positions", "out_loc"))
    eng.attn_backend.prepare_metadata(batch)
    with eng.ctx.forward_batch(batch):
        use_graph = eng.graph_runner.can_use_cuda_graph(batch)
        assert use_graph == f["graph"], "the two engines disagree on graph replay for this batch"
        logits = eng.graph_runner.replay(batch) if use_graph else eng.model.forward(eng.ctx, batch)
    return logits[: f["size"]]
