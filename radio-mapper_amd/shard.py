"""Multi-GPU decomposition of the path: capture windows (and frequency channels) are independent
(tdoa_processor.py:363 loops frequency groups independently; no state crosses windows), so ranks get
contiguous blocks of windows and there is NO collective on the data path.  The only exchange is a
host-side gather of the 12-byte-per-pair-window results."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def window_shard(n_windows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [start, start+count) of rank `rank`; remainders go to the low ranks."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_windows, world_size)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def gather_lags(lag_int: np.ndarray, lag_frac: np.ndarray, peak: np.ndarray, dst: int = 0
                ) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """Concatenate every rank's [W_r][P] results in rank order on rank `dst` (None elsewhere).
    Uses torch.distributed's object gather (host side; works under gloo and nccl)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return lag_int, lag_frac, peak
    world = dist.get_world_size()
    payload = (np.ascontiguousarray(lag_int), np.ascontiguousarray(lag_frac), np.ascontiguousarray(peak))
    bucket: Optional[List] = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(payload, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    return tuple(np.concatenate([b[k] for b in bucket], axis=0) for k in range(3))
