"""Chain sharding over the GPUs of one node and the one exchange step of the path.

Chains are independent between refits (core/sample.py:211-213 is a pure map), so each rank owns a contiguous
block of chains and runs them with no data-path collective; the per-chain RNG stream is keyed by the GLOBAL
chain index, so results do not depend on the number of ranks.  The refit needs every rank to see all samples
(the reference's resampler indexes the flattened all-chain array, core/recipe.py:1024-1025,1074,1082): that is
ONE all-gather per sampling round (RCCL over xGMI with backend "nccl"; gloo in the CPU tests)."""
import numpy as np

__all__ = ['world', 'shard_range', 'all_gather_chains', 'local_device', 'broadcast_int', 'all_reduce_sum', 'all_gather_stack']


def world():
    """(rank, world_size) of the default process group, (0, 1) without one."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def local_device(n_visible, env=None):
    """Device index this rank should use on its node: one process per GPU (torchrun exports LOCAL_RANK).  Returns None
    when there is nothing to decide (single process, or no LOCAL_RANK)."""
    import os
    env = os.environ if env is None else env
    rank, ws = world()
    if ws == 1 or 'LOCAL_RANK' not in env:
        return None
    lr = int(env['LOCAL_RANK'])
    if env.get('BFHIP_SHARE_DEVICE') and n_visible > 0:
        return lr % n_visible  # several ranks per GPU, on request (the two-rank tests on a one-GPU box)
    if not 0 <= lr < n_visible:
        raise RuntimeError('LOCAL_RANK {} but only {} visible device(s).'.format(lr, n_visible))
    return lr


def shard_range(n_chain, rank, world_size):
    """Contiguous block [begin, end) of chains owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(int(n_chain), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_gather_chains(t, n_chain):
    """All-gather along the chain axis of a tensor whose first dimension is this rank's shard.

    Shards may differ by one chain: every rank pads to the largest shard, one ``all_gather`` moves the padded
    blocks, and the pads are cut away.  Works on CUDA tensors (nccl = RCCL) and CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return t
    sizes = [shard_range(n_chain, r, ws)[1] - shard_range(n_chain, r, ws)[0] for r in range(ws)]
    mx = max(sizes)
    pad = t
    if t.shape[0] < mx:
        pad = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))], 0)
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)


def broadcast_int(v, src=0):
    """The value rank ``src`` holds, on every rank (a no-op without a process group)."""
    rank, ws = world()
    if ws == 1:
        return int(v)
    import torch
    import torch.distributed as dist
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src)
    return int(t.item())


def all_reduce_sum(t):
    """Sum of a tensor over the ranks, in place (a no-op without a process group)."""
    rank, ws = world()
    if ws > 1:
        import torch.distributed as dist
        if dist.get_backend() == 'gloo' and t.is_cuda:  # (plumbing tests: gloo moves host tensors)
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_stack(t):
    """(world_size, *t.shape): the tensor ``t`` (same shape on every rank) of every rank, on every rank."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return t[None]
    stage = dist.get_backend() == 'gloo' and t.is_cuda  # (plumbing tests: gloo moves host tensors)
    src = t.cpu().contiguous() if stage else t.contiguous()
    out = [torch.empty_like(src) for _ in range(ws)]
    dist.all_gather(out, src)
    res = torch.stack(out)
    return res.to(t.device) if stage else res
