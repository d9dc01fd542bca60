"""Frame sharding and the super-frame index exchange.

Frames are independent (the encoder never writes frame_seq, OLD:1142-1150; contexts are per-call constants), so they are
sharded round-robin over ranks with NO data-path collective.  The only exchange step is one all-gather of fixed-size
per-frame index records (t3_frame_record, 96 B: frame index, word count, CRC-32, symbol sum, the 54 header symbols)
from which every rank assembles the T3V-style frame index (offset, words; io_t3p_t3v.cpp:252-289).

The exchange itself is the library's: t3hip_index_allgather = ncclAllGather (RCCL over xGMI) on a communicator the host
creates with t3hip_comm_unique_id / t3hip_comm_create — the same calls a C++ host makes.  This module only
carries the 128-byte rendezvous id between the ranks, over whatever process group the caller already has
(torch.distributed here).  `gather_records_torch` is the CPU rehearsal of the same control flow over gloo (tests only)."""
import numpy as np

from . import Comm, FRAME_RECORD_BYTES, PAD_FRAME_IDX, comm_available, comm_unique_id, index_assemble


def frames_of_rank(n_frames, rank, world):
    """Round-robin shard: frame f lives on rank f % world."""
    return list(range(rank, n_frames, world))


def make_comm(group=None):
    """RCCL communicator over the ranks of a torch.distributed group, or None -- on EVERY rank -- when any rank cannot have one.
    The rendezvous cannot strand a rank: (1) every rank probes the binding locally (t3hip_comm_available: dlopen only, no
    collective), (2) the ranks agree with an all-reduce(MIN) on the group's CPU backend, (3) rank 0 draws the unique id only if all
    ranks are able, and the broadcast of (status byte, id) happens on every path, so the collectives match whatever failed,
    (4) only then does anyone enter ncclCommInitRank.  The group needs a CPU-capable backend (e.g. "cpu:gloo,cuda:nccl").  Call
    after t3.init(local_device)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    able = torch.tensor([1 if comm_available() else 0])
    dist.all_reduce(able, op=dist.ReduceOp.MIN, group=group)
    box = torch.zeros(129, dtype=torch.uint8)
    if rank == 0 and int(able.item()) == 1:
        try:
            box[1:].copy_(torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8)); box[0] = 1
        except Exception:   # noqa: BLE001  (ncclGetUniqueId failed: the others learn it from the status byte)
            box[0] = 0
    dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if int(box[0].item()) != 1:
        return None
    return Comm(bytes(box[1:].numpy().tobytes()), world, rank)


def gather_records(comm, local_records, stream=None):
    """local_records: uint8 CUDA tensor [n_local, 96]; every rank passes the same n_local (pad with zero records whose
    frame_idx == PAD_FRAME_IDX).  Returns the [world * n_local, 96] tensor, rank-major; asynchronous on `stream`
    (default: torch's current stream)."""
    import torch
    assert local_records.is_cuda and local_records.dtype == torch.uint8 and local_records.shape[1] == FRAME_RECORD_BYTES
    local_records = local_records.contiguous()
    out = torch.empty((comm.world * local_records.shape[0], FRAME_RECORD_BYTES), dtype=torch.uint8, device=local_records.device)
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    comm.index_allgather(local_records.data_ptr(), local_records.shape[0], out.data_ptr(), s)
    return out


def gather_records_torch(local_records, group=None):
    """The same exchange through torch.distributed.all_gather_into_tensor (gloo on CPU tensors): rehearsal of the N>1 control
    flow where no RCCL communicator can exist (CPU tests; two ranks sharing one card)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * local_records.shape[0], FRAME_RECORD_BYTES), dtype=torch.uint8, device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)
    return out


def assemble_index(gathered, first_payload_offset=0):
    """Sort by frame index, drop padding records, prefix-sum the payload offsets."""
    recs = index_assemble(np.asarray(gathered.cpu() if hasattr(gathered, "cpu") else gathered, dtype=np.uint8).reshape(-1), first_payload_offset)
    return [r for r in recs if r.frame_idx != PAD_FRAME_IDX]
