"""One-process-per-GPU driver of the sharded ICP loop.

The reference is single-process; this is the one exchange step the data-parallel design adds:
scene (source) points are split into contiguous shards, one per rank, the model (target) index is
replicated, and every iteration all-reduces the 17 fp64 sums {n, Σs, Σt, Σ t sᵀ, Σd²} so that each
rank computes the same incremental transform redundantly (no broadcast, no other data-path
collective).  The collective goes through torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests); the library's own RCCL communicator (ope_comm_init_rank) is the
alternative that keeps the whole loop in C++.

`run_sharded_icp` only needs an *engine* with
    begin(), accumulate(), sums (a torch tensor of 17 float64), update(), poll() -> result, end() -> result
so the CPU tests can drive exactly this loop with a checker engine and `gloo`.
"""
from __future__ import annotations

from dataclasses import dataclass


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for `rank` of `world`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * n // world, (rank + 1) * n // world


@dataclass
class ShardedResult:
    T: object
    iterations: int
    converged: bool
    state: int
    last_mse: float
    n_corr: int
    align_strength: float


class GpuEngine:
    """The HIP path: kernels via the C ABI on torch's current stream, sums in a CUDA tensor."""

    def __init__(self, ope, ctx, src_cloud, tgt_index, params, guess=None, n_src_total=None, n_tgt_total=None):
        import torch
        self.ope, self.ctx = ope, ctx
        self.src, self.tgt, self.params, self.guess = src_cloud, tgt_index, params, guess
        self.n_src_total, self.n_tgt_total = n_src_total, n_tgt_total
        self.sums = torch.zeros(ope.NUM_SUMS_MAX, dtype=torch.float64, device=f"cuda:{ctx.device}")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.icp_set_sums_buffer(self.sums.data_ptr())

    def begin(self):
        if self.n_src_total is not None:
            self.ctx.icp_set_global_sizes(self.n_src_total, self.n_tgt_total)
        self.ctx.icp_begin(self.src, self.tgt, self.params, self.guess)

    def accumulate(self):
        self.ctx.icp_accumulate()

    def update(self):
        self.ctx.icp_update()

    def poll(self):
        return self.ctx.icp_poll()

    def end(self):
        out = self.ctx.icp_end()
        self.ctx.icp_set_sums_buffer(None)
        return out


def run_sharded_icp(engine, max_iterations: int, check_every: int = 10, group=None) -> ShardedResult:
    """accumulate -> all-reduce(17 x fp64) -> update, polling the convergence flag every `check_every`."""
    import torch.distributed as dist

    engine.begin()
    it = 0
    max_iterations = max(int(max_iterations), 1)
    while it < max_iterations:
        batch = min(check_every, max_iterations - it) if check_every > 0 else max_iterations - it
        for _ in range(batch):
            engine.accumulate()
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                dist.all_reduce(engine.sums, op=dist.ReduceOp.SUM, group=group)
            engine.update()
        it += batch
        if it < max_iterations:
            r = engine.poll()
            # every rank holds the same replicated state, so this decision is identical everywhere
            if r.converged or r.state == 5:
                break
    out = engine.end()
    return ShardedResult(out.T, out.iterations, bool(out.converged), out.state, out.last_mse, int(out.n_corr),
                         out.align_strength)
