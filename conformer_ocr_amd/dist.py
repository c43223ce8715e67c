"""Multi-GPU plumbing of the recognition path.

Lines are independent units: every rank holds a full weight replica and takes its own batches; the
only collective on the weights is ONE broadcast of the packed blob at start-up (RCCL over xGMI via
torch.distributed's "nccl" backend); results come back as small label records.  There is no
collective in the steady-state loop."""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def broadcast_weights(engine, src: int = 0, group=None, source=None) -> None:
    """Rank `src` ran `finalize()`, the others `finalize_empty()`: ONE broadcast of the packed device blob
    (RCCL over xGMI with the "nccl" backend), through a torch-owned staging buffer (`cocr_blob_export` / `_import`).
    The blob holds the weights only (43 MB for the 12-block D=256 model); every rank derives the positional tables
    (61 MB) from the broadcast pos_proj matrices on its own device at its next forward.
    `source`: the engine that holds the weights on rank `src` if it is not `engine` itself (several engines per rank;
    a single-rank rehearsal of the receive path) -- `engine` then imports the broadcast blob on EVERY rank."""
    holder = engine if source is None else source
    buf = torch.empty(engine.blob_nbytes(), dtype=torch.uint8, device=engine.device)
    if dist.get_rank(group) == src:
        holder.export_blob(buf)
    dist.broadcast(buf, src=src, group=group)
    if dist.get_rank(group) != src or holder is not engine:
        engine.import_blob(buf)
    torch.cuda.synchronize(engine.device)


def broadcast_state_dict(state: Optional[Mapping[str, np.ndarray]], names_shapes: Mapping[str, Sequence[int]],
                         src: int = 0, group=None) -> Dict[str, np.ndarray]:
    """Host-side alternative (any backend, e.g. gloo): broadcast the fp32 state dict as one flat tensor.
    `names_shapes` (from spec.model_state_spec) fixes order and sizes on every rank."""
    sizes = [int(np.prod(s)) if len(s) else 1 for s in names_shapes.values()]
    flat = torch.empty(sum(sizes), dtype=torch.float32)
    if dist.get_rank(group) == src:
        flat.copy_(torch.from_numpy(np.concatenate([np.asarray(state[k], dtype=np.float32).reshape(-1) for k in names_shapes])))
    dist.broadcast(flat, src=src, group=group)
    out, off = {}, 0
    a = flat.numpy()
    for (k, shp), n in zip(names_shapes.items(), sizes):
        out[k] = a[off:off + n].reshape(tuple(shp)).copy()
        off += n
    return out


def shard_batches(n_batches: int, rank: int, world: int) -> List[int]:
    """Round-robin partition of a queue of width-bucketed batches: rank r takes r, r+world, ..."""
    return list(range(rank, n_batches, world))


def bucket_width(width: int, edge: int = 200) -> int:
    """Fixed bucket edges: a line is padded to the next multiple of `edge`, never to its batch's maximum,
    so that a line's logits (which depend on the padded width: SURVEY 0.6) do not depend on which other
    lines or how many ranks there are."""
    return -(-int(width) // edge) * edge


def gather_strings(local: Sequence[str], local_ids: Sequence[int], total: int, group=None) -> Optional[List[str]]:
    """Collect per-rank results on rank 0 in global line order (reporting path, not the hot path)."""
    world = dist.get_world_size(group)
    objs: List = [None] * world
    dist.all_gather_object(objs, (list(local_ids), list(local)), group=group)
    if dist.get_rank(group) != 0:
        return None
    out: List[Optional[str]] = [None] * total
    for ids, strs in objs:
        for i, s in zip(ids, strs):
            out[i] = s
    return out  # type: ignore[return-value]
