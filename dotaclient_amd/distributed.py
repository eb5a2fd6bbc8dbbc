"""Data-parallel gradient averaging over RCCL (one process per GPU).

Replaces /root/reference/distributed.py:16-79 (`DistributedDataParallelSparseParamCPU`): the
reference broadcasts the 34 parameters from rank 0 (:71-74) and, after backward, issues 68 blocking
gloo collectives (per parameter: an int64 has-grad count + the gradient, :29-57).  Here the whole
gradient lives in ONE flat fp32 bucket (3.06 MB for the reference network) with the five per-head
has-grad flags appended, so a step costs a single all-reduce(SUM) over xGMI plus one scaling kernel
(dc_dp_average_grads) that divides every parameter by the number of ranks that had a gradient for it.
Semantics kept from the reference: each rank normalises advantages and takes loss means over ITS OWN
shard (optimizer.py:588), gradients are AVERAGED over the ranks that have one, and a rank whose head
never acted keeps "no gradient" for that head (its Adam skips it).
"""
import torch
import torch.distributed as dist

from . import _lib


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class FlatGradAllReducer:
    """overlap (default on; DC_DP_OVERLAP=0 or overlap=False for the single collective): the bucket is reduced in two
    collectives.  The gradients of every parameter from affine_pre_rnn on (82 % of the bucket, final once the first half of the
    backward is enqueued) plus the head flags go first, asynchronously, while the embedding backward runs; the embedding
    gradients follow.  Same sums, same averaging: both forms are checked against the reference wrapper's two-rank fixture
    with real engines (tests/test_gpu_dp.py)."""

    def __init__(self, engine, group=None, overlap=None):
        import os
        self.engine = engine
        self.group = group
        self.overlap = (os.environ.get('DC_DP_OVERLAP', '1') != '0') if overlap is None else bool(overlap)
        self._work = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = engine.device
        # bucket = [flat grads | 5 head flags | world] so that one collective carries everything
        self.bucket = torch.zeros(engine.total + 8, dtype=torch.float32, device=dev)
        engine.grads = self.bucket[:engine.total]
        self.tail = self.bucket[engine.total:]

    def sync_parameters(self):
        """distributed.py:71-74: every rank starts from rank 0's weights (one broadcast of the flat buffer)."""
        if self.world > 1:
            dist.broadcast(self.engine.params, src=0, group=self.group)
            self.engine.params_changed()

    def _set_tail(self, engine):
        self.tail[:5] = engine.head_on[:5].to(torch.float32)
        self.tail[5] = 1.0

    def start_upper(self, engine):
        """After the first half of the backward (Engine.backward(.., DC_DIMS_BWD_UPPER)) has been enqueued."""
        if self.world <= 1:
            return
        self._set_tail(engine)
        self._work = dist.all_reduce(self.bucket[engine.embed_floats:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, engine):
        """After the second half: reduce the embedding gradients, join the first collective, average."""
        if self.world <= 1:
            return
        dist.all_reduce(self.bucket[:engine.embed_floats], op=dist.ReduceOp.SUM, group=self.group)
        self._work.wait()
        self._work = None
        self._average(engine)

    def __call__(self, engine):
        """grad_hook for Engine.train_epoch: runs between backward and the Adam step."""
        if self.world <= 1:
            return
        self._set_tail(engine)
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        self._average(engine)

    def _average(self, engine):
        _lib.check(engine.lib.dc_dp_average_grads(
            _lib.ptr(engine.seg_off), _lib.ptr(engine.seg_len), _lib.ptr(engine.seg_gate), len(engine.seg_names),
            engine.max_seg_len, _lib.ptr(engine.grads), _lib.ptr(self.tail), 0.0, _lib.stream_ptr()),
            'dc_dp_average_grads')
