"""Frame sharding and the super-frame index exchange.

Frames are independent (the encoder never writes frame_seq, OLD:1142-1150; contexts are per-call constants), so they are
sharded round-robin over ranks with NO data-path collective.  The only exchange step is one all-gather of fixed-size
per-frame index records (t3_frame_record, 96 B: frame index, word count, CRC-32, symbol sum, the 54 header symbols)
from which every rank assembles the T3V-style frame index (offset, words; io_t3p_t3v.cpp:252-289).
backend "nccl" = RCCL over xGMI on the GPU node; "gloo" in the CPU tests."""
import numpy as np

from . import FRAME_RECORD_BYTES, index_assemble


def frames_of_rank(n_frames, rank, world):
    """Round-robin shard: frame f lives on rank f % world."""
    return list(range(rank, n_frames, world))


def gather_records(local_records, group=None):
    """local_records: uint8 tensor [n_local, 96] (device tensor for nccl, CPU tensor for gloo); every rank must pass the
    same n_local (pad with zero records whose n_words == 0 and frame_idx == 2**64-1).  Returns [world * n_local, 96]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * local_records.shape[0], FRAME_RECORD_BYTES), dtype=torch.uint8, device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)
    return out


def assemble_index(gathered, first_payload_offset=0):
    """Sort by frame index, drop padding records, prefix-sum the payload offsets."""
    recs = index_assemble(np.asarray(gathered.cpu() if hasattr(gathered, "cpu") else gathered, dtype=np.uint8).reshape(-1), first_payload_offset)
    return [r for r in recs if r.frame_idx != 2**64 - 1]


PAD_FRAME_IDX = 2**64 - 1
